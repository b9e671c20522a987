// kernels.h — launchers of the gfx950 kernels, called by the C-ABI layer (engine.hip).
// Every launcher enqueues on `stream` and returns the hipError_t of the launch; none synchronises.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/slam_hip.h"

namespace slam {

// Optional timing of ONE kernel: when non-null, `start` is recorded immediately before and `stop`
// immediately after that kernel's launch on the same stream.
struct EventPair {
    hipEvent_t start = nullptr, stop = nullptr;
};

// ---- score_kernels.hip (reference: Subsystem_1/main.c:381-596)
struct ScoreGrid {
    const float* edt;   // [rows][ld]
    int rows, cols, ld;
    float ipix;         // 1 / pixel, computed on the host with one float division (main.c:383)
    float min_x, min_y;
    // Optional packed copy of the same grid for the many-pose scorers (launch_edt_pack): one BYTE per cell, stored in
    // vertical strips of 16 columns — cell (ix, iy) at (ix >> 4) * strip_bytes + iy * 16 + (ix & 15) — so that a 128-byte
    // line holds a 16 x 8 patch of cells, and a 256-entry table that turns a byte back into the cell's float, bit for bit.
    const uint8_t* packed = nullptr;
    const float* table = nullptr;
    int strip_bytes = 0;   // 16 * (rows rounded up to 8)
};
// The capped EDT (main.c:223-269, Appendix A.4) holds few distinct values: 0 on an occupied cell, sqrtf(d2) of a small
// integer d2 below the cap, and the cap itself.  launch_edt_pack writes code(cell) = d2 (255: the grid's largest value) into
// `packed`, table[code] = the float it stands for, and flag[1] = 1 if some cell of the rows x cols rectangle is not table[its
// code] bit for bit (a grid that did not come from the EDT kernels, or a cap above 15.9 cells): the caller then keeps the
// float grid.  flag: two device words {largest value's bits (scratch), bad}, zeroed by the launcher.
size_t edt_packed_bytes(int rows, int cols);
hipError_t launch_edt_pack(hipStream_t stream, const float* edt, int ld, int rows, int cols, uint8_t* packed, float* table,
                           uint32_t* flag);
hipError_t launch_score_poses(hipStream_t stream, const ScoreGrid& g, const float* bx, const float* by, int nbeams,
                              const float* x, const float* y, const float* th_or_ct, const float* st_or_null,
                              int nposes, float* score, int32_t* count, const EventPair* ev = nullptr);
// motion sample fused in front of the score: pose' = motion(src[anc]), written to dst and scored
// A launch_free_list (paged_kernels.hip: the free list of a paged session, its arguments below) that travels in workgroups of its
// own behind those of a motion + score launch: neither needs anything from the other (free_list_body.h).
struct FreeListRider {
    const uint32_t* stamp = nullptr;   // nullptr: no rider
    int npages = 0;
    uint32_t live = 0;
    int32_t* freelist = nullptr;
    int32_t* pool_state = nullptr;
    int32_t* h_short = nullptr;
    int first_block = 0, nblocks = 0;   // filled in by the launcher
};
struct MotionIO {
    const float *sx, *sy, *sth;
    const int32_t* anc;
    float *x, *y, *th;
    FreeListRider rider;
};
// rider (optional): *rode says whether the launch took it along (the one-wavefront-per-pose form of small batches does not:
// the caller then launches the list by itself)
hipError_t launch_motion_score(hipStream_t stream, const ScoreGrid& g, const float* bx, const float* by, int nbeams,
                               const MotionIO& io, int nposes, int64_t first_id, const float dp[3], const float sigma[3],
                               uint64_t seed, uint32_t frame, float* score, int32_t* count,
                               const EventPair* ev = nullptr, bool* rode = nullptr);
int free_list_blocks(int npages);
hipError_t launch_pose_hits(hipStream_t stream, const ScoreGrid& g, const float* bx, const float* by, int nbeams,
                            const float* pose_xycs /*4 floats on device*/, float* hits, int32_t* count);

// The reference's 27-candidate lattice in one launch (FastMatch, main.c:440-573): candidate c gets one
// wavefront; score[c], count[c] and, in merged_hits, exactly what the reference leaves in its shared
// bestHits[] scratch after the sweep — entry j holds the hit of the LAST candidate (in evaluation
// order) that had more than j in-bounds beams (main.c:515 overwrites the prefix for every candidate).
// work: device scratch of 27*nbeams floats.  out layout: score[27] | count[27] (int) | maxcount (int) | merged[nbeams]
// d_nbeams (optional): the beam count lives on the device (<= nbeams, which then is the capacity / row stride).
// persist (optional): device mirror of the caller's persistent hit scratch, updated like the host copy.
// host_out / host_flag / seq (optional): also deliver `out` to mapped pinned host memory and release `seq`.
hipError_t launch_lattice(hipStream_t stream, const ScoreGrid& g, const float* bx, const float* by, int nbeams,
                          const int32_t* d_nbeams, const float* cand_xycs /* X[27] Y[27] CT[27] ST[27] */, float* work,
                          float* out, float* persist, float* host_out, uint32_t* host_flag, uint32_t seq);
// FastMatch on g1 followed by FastMatch2 on g2 around its winner, THREE launches and no host in between (engine.hip:
// slam_engine_fastmatch_pair): the first lattice from cand1 as above; the second lattice's wavefronts work out the first call's
// winner themselves (strict '<' from +inf over out1's 27 scores, the first of equals; none: the middle candidate = the input
// pose) and lay their own candidate around it — x, y = the winner's -/+ pair_in[18], heading (cos, sin) = pair_in[3 a + b],
// pair_in[9 + 3 a + b] for the winner's heading a and the candidate's b —; one merge for both calls: the persistent hit scratch
// as the two calls one after the other leave it (entry j: the second call's where it has one, else the first call's), scores
// and counts of both calls to the host (host_out1: 54 words; host_out2: scores, counts, the second call's maxcount), the flag.
hipError_t launch_lattice_pair(hipStream_t stream, const ScoreGrid& g1, const ScoreGrid& g2, const float* bx, const float* by, int nbeams,
                               const int32_t* d_nbeams, const float* cand1, const float* pair_in, float* work1, float* work2, float* out1,
                               float* out2, float* persist, float* host_out1, float* host_out2, uint32_t* host_flag, uint32_t seq);

// ---- mapper_kernels.hip (reference: main.c:71-198, 271-354, 941-953)
hipError_t launch_clean_scan(hipStream_t s, const float* range, const float* cos_tab, const float* sin_tab, int nbeams,
                             float range_min, float usable, float* bx, float* by, int32_t* nscan);
hipError_t launch_transform(hipStream_t s, const float* bx, const float* by, const int32_t* nscan, float px, float py,
                            float ct, float st, float* tx, float* ty);
hipError_t launch_crop(hipStream_t s, const float* tx, const float* ty, const int32_t* nscan, float border,
                       const float* mx, const float* my, const int32_t* msize, int local_cap, float* lx, float* ly,
                       int32_t* lsize);
hipError_t launch_rasterise(hipStream_t s, const float* lx, const float* ly, const int32_t* lsize, float pixel, int ld,
                            int32_t* grid, slam_grid_meta* meta);
hipError_t launch_map_append(hipStream_t s, const float* hits, int nhits, const float* tx, const float* ty, float* mx,
                             float* my, int32_t* msize, int map_cap, float threshold);

// ---- edt_kernels.hip (row A6; reference: main.c:223-269, main_accelerated.c:215-283)
enum { EDT_MAX_RADIUS = 32 };
hipError_t launch_edt(hipStream_t stream, const int32_t* occ, int ld, int rows, int cols, float cap, float* out,
                      const EventPair* ev = nullptr);

// ---- pf_kernels.hip (rows A9-A12; no reference counterpart)
hipError_t launch_motion_sample(hipStream_t stream, const float* sx, const float* sy, const float* sth,
                                const int32_t* anc, float* x, float* y, float* th, int n, int64_t first_id,
                                const float dp[3], const float sigma[3], uint64_t seed, uint32_t frame);

struct EkfArgs {
    const float* map_in;
    float* map_out;
    int64_t row_stride;    // floats between consecutive particles' rows
    int plane_stride;      // floats between the five planes inside a row (>= nlandmarks)
    int nlandmarks;
    const float *x, *y, *th;
    const int32_t* anc;
    int n;
    const float *obs_zx, *obs_zy;   // device, indexed by landmark, padded to an even length; zx NaN = not observed
    float meas_var;
    float* loglik;        // [n], always written
    float* loglik_user;   // optional second copy for the caller
    int xcd_chunk;        // set by the launcher: workgroups per XCD when the grid is renumbered XCD-contiguously, else 0
    // ---- split layout (cov != nullptr; split_kernels.hip): map_in / map_out hold the MEANS only — rows of two planes
    // (mu_x, mu_y) of plane_stride floats, row_stride floats apart — and the covariance planes (P_xx, P_xy, P_yy) exist once
    // per COVARIANCE CLASS: in the world-frame update the posterior covariance of a landmark depends on its prior covariance,
    // on q and on whether the frame observes it, never on the particle, so particles whose covariances were equal once stay
    // equal for ever and share one row cov[class].  The update reads it (launch_cov_update rewrites it afterwards, once per
    // class), hands the class of the source particle on to its offspring and marks it as still in use.
    const float* cov = nullptr;       // [classes][3][plane_stride]
    int64_t cov_stride = 0;           // floats between the rows of two classes
    const float* covx = nullptr;      // [classes][2][plane_stride]: 1 / det (P + q I) and 0.5 log det (P + q I) of the same covariances
    int64_t covx_stride = 0;          //   (ekf_math.h: ekf_det_terms), kept up to date by whoever writes cov
    const int32_t* cls_in = nullptr;  // class of every source particle
    int32_t* cls_out = nullptr;       // class of every particle of this frame
    uint32_t* cstamp = nullptr;       // [classes]: stamp_now = "a particle of this frame belongs to the class"
    uint32_t stamp_now = 0;
    // Sharded sessions, split layout: a frame's update in two launches.  Source rows below n are this rank's own particles;
    // rows from n on are the staging tail, filled by the exchange of the frame.  group_filter 1: only the groups (the particles
    // of one wavefront) whose sources are ALL local — they need nothing from the exchange and go out with the score, in the
    // fused front launch, before the host has even looked at the exchange plan; 2: only the other groups, behind the unpack.
    // 0: every group.
    int group_filter = 0;
};
// the split layout's part of EkfArgs, as the session hands it to the engine's stage functions
struct SplitIO {
    int group_filter;          // EkfArgs::group_filter
    const int32_t* map_anc;    // the gather index of the MAPS when it is not the scorer's pose index (sharded: poses come out of
                               // the all-gathered array of every rank, maps out of the local rows + staging tail); nullptr: the same
    const float* cov;
    int64_t cov_stride;
    const float* covx;
    int64_t covx_stride;
    const int32_t* cls_in;
    int32_t* cls_out;
    uint32_t* cstamp;
    uint32_t stamp_now;
};
// group_size: 0 = one wavefront per particle; 2 / 4 / 8 = the grouped out-of-place form (that many neighbouring particles
// per wavefront share their source rows in registers) — a speed choice only, every form gives the same bits
hipError_t launch_ekf_update(hipStream_t stream, const EkfArgs& a, const EventPair* ev = nullptr, int group_size = 0);
// motion sample + scan-match score AND the grouped out-of-place landmark update in ONE launch (single-GPU frames on rows): the
// gathers of the scorer run in the shadow of the update's row stores.  `a.x / a.y / a.th` are not read (the update works out
// its particles' motion samples itself, the same bits the scorer writes to io.x / io.y / io.th).  *launched = false: shapes
// that this kernel does not take; nothing was issued.
hipError_t launch_selftest_reciprocal(hipStream_t stream, unsigned long long* out /* [2], zeroed: mismatches, values checked */);
bool frame_front_fits(int n, int nlandmarks, int group_size);   // the shapes launch_frame_front takes
hipError_t launch_frame_front(hipStream_t stream, const ScoreGrid& g, const float* bx, const float* by, int nbeams,
                              const MotionIO& io, int64_t first_id, const float dp[3], const float sigma[3], uint64_t seed,
                              uint32_t frame, float* score, int32_t* count, const EkfArgs& a, int group_size,
                              const EventPair* ev, bool* launched, int* lanes_per_pose = nullptr);

// ---- split_kernels.hip: the covariance classes of the split layout (see EkfArgs)
// The posterior covariance of every class that is still in use, in place, once per class: P' = (I - W) P for an observed
// landmark seen before, q I for a first sighting, the prior without an observation — the values every particle of the class
// would have written into its own row (ekf_math.h: ekf_shared).  `live`: the classes that were in use after the previous
// frame, cnt[phase] of them; those that still are (cstamp == stamp_now) are updated and appended to the next list (cnt[(phase
// + 1) % 3]; cnt[(phase + 2) % 3] is zeroed for the frame after).  bound >= cnt[phase]: the launch's width (the host's stale
// knowledge is good enough: the list only shrinks).  h_live (mapped host memory): {cnt[phase], epoch}.
struct CovArgs {
    float* cov;
    int64_t cov_stride;
    float* covx;           // the determinant terms of the NEW covariances go here (EkfArgs::covx)
    int64_t covx_stride;
    int plane_stride, nlandmarks;
    const float *obs_zx, *obs_zy;
    float meas_var;
    const int32_t* live_in;
    int32_t* live_out;
    int32_t* cnt;          // [3]
    int phase;
    const uint32_t* cstamp;
    uint32_t stamp_now;
    int32_t* h_live;       // 8-byte word {cnt[phase], epoch} in mapped host memory
    int32_t* h_mark;       // ... and, written behind a system-scope fence, {mark, epoch}
    uint32_t epoch;
    uint32_t mark;         // the host's running count of classes appended to the list so far (sharded sessions: rows received)
};
hipError_t launch_cov_update(hipStream_t stream, const CovArgs& a, int bound, const EventPair* ev = nullptr);
// rows [n][5][plane_stride_in] (row_stride_in floats apart) -> means [n][2][Lp], classes, class rows [..][3][Lp]: neighbouring
// particles whose three covariance planes are equal bit for bit share a class (classes are numbered 0, 1, .. in particle
// order; all particles alike -> one class).  scratch: split_scratch_words(n) int32 words.  live[k] = k, cnt[phase] = number
// of classes (cnt[other] = 0), every class stamped stamp_now; h_live = {classes, epoch}.
size_t split_scratch_words(int n);
// covx (written by a second launch, so that it may lie where the rows came from): the determinant terms of every class.
hipError_t launch_split_from_rows(hipStream_t stream, const float* rows, int64_t row_stride_in, int plane_stride_in, int nlandmarks,
                                  int n, int Lp, float* mean, float* cov, float* covx, float meas_var, int32_t* cls, int32_t* live,
                                  int32_t* cnt, int phase, uint32_t* cstamp, uint32_t stamp_now, int32_t* h_live, uint32_t epoch,
                                  void* scratch);
// out row k = [means of particle idx[k] (or k) | the covariance planes of its class], nlandmarks columns of each plane
hipError_t launch_rows_from_split(hipStream_t stream, const float* mean, const float* cov, const int32_t* cls, int Lp,
                                  const int32_t* idx, int count, float* rows, int64_t row_stride, int plane_stride, int nlandmarks);
// Sharded sessions on the split layout.  A migrating particle travels as the same record whatever the layouts of the two ranks
// (pose, then five planes of nlandmarks values): launch_migrate_pack(.., split_cov, split_cls) reads it from the means and the
// class's covariance row, and this launch puts record p of `in` into staging row n + p of the means, with a class of its own —
// first_class + p, covariance planes and determinant terms filled in, appended to the list of classes in use (live[*cnt ..)).
// Class numbers for the arrivals come from a free list on the device: the classes no current particle belongs to, i.e. whose
// stamp is older than `min_live` — the stamp of the last update whose particles are the current ones.  At most n classes are in
// use and there are n + staging rows of them, so a list made anew holds at least as many numbers as a rank has staging rows,
// whatever else it holds: the HOST hands them out (entries [first, first + total) of the list go to this launch) and asks for
// a new list (launch_class_free_list, in front of the unpack) when the guaranteed part of the old one is used up.
//   fs: two int32 words on the device, zero before the first launch and zero again behind every launch.
hipError_t launch_class_free_list(hipStream_t stream, const uint32_t* cstamp, int nclasses, uint32_t min_live, int32_t* freelist,
                                  int32_t* fs);
// the arrivals' classes are stamped `stamp` (= min_live: in use until the next update has said which of them have offspring)
hipError_t launch_migrate_unpack_split(hipStream_t stream, const float* in, int total, int n, float* pose, int64_t pose_ld, float* mean,
                                       float* cov, float* covx, int32_t* cls, int Lp, int nlandmarks, float meas_var,
                                       const int32_t* freelist, int first, uint32_t* cstamp, uint32_t stamp, int32_t* live,
                                       int32_t* cnt);
// a frame without a landmark update: means and classes follow their particles (out[i] = in[anc[i]])
hipError_t launch_split_gather(hipStream_t stream, const float* mean_in, float* mean_out, const int32_t* cls_in, int32_t* cls_out,
                               int Lp, const int32_t* anc, int n, uint32_t* cstamp, uint32_t stamp_now);
// the classes alone follow their particles (split pages: the means follow through the page tables)
hipError_t launch_class_gather(hipStream_t stream, const int32_t* cls_in, int32_t* cls_out, const int32_t* anc, int n, uint32_t* cstamp,
                               uint32_t stamp_now);
// every particle: all landmarks "not seen yet", one class
hipError_t launch_split_reset(hipStream_t stream, float* mean, float* cov, float* covx, int32_t* cls, int Lp, int n, int32_t* live,
                              int32_t* cnt, int phase, uint32_t* cstamp, uint32_t stamp_now, int32_t* h_live, uint32_t epoch);

// carry / prev_resampled (optional): see logweight_kernel — the weights a frame without resample left behind
// In-place update of the OBSERVED landmarks only (frames that keep their population): the observation table is first
// compacted into a list in landmark order (ids, measurements, accumulator rounds; count[2] = {nobs, highest round} on the
// device, {nobs, L} optionally in mapped host memory), then one lane per observation gathers, updates and scatters.
constexpr int kObsListMaxLandmarks = 65536;   // the list form keeps a bitmap of the observed landmarks in LDS
hipError_t launch_build_obs_list(hipStream_t stream, const float* tzx, const float* tzy, int L, int32_t* id, float* zx,
                                 float* zy, int32_t* round, int32_t* count, int32_t* h_count);
hipError_t launch_ekf_sparse(hipStream_t stream, const EkfArgs& a, const int32_t* id, const float* zx, const float* zy,
                             const int32_t* round, const int32_t* count, const EventPair* ev = nullptr);
// ---- paged_kernels.hip: landmark maps as copy-on-write pages of kPageLandmarks landmarks (5 planes x 32 floats = 640 B)
// (SLAM_PAGE_LANDMARKS: a power of two <= 32, for measurement builds — profiles/collect_page_sizes.sh; the product is built with 32)
#ifndef SLAM_PAGE_LANDMARKS
#define SLAM_PAGE_LANDMARKS 32
#endif
constexpr int kPageLandmarks = SLAM_PAGE_LANDMARKS;
// the compact observation list launch_build_obs_list makes: ids ascending, measurements, accumulator rounds, {nobs, highest round}
struct ObsListView {
    const int32_t* id = nullptr;
    const float *zx = nullptr, *zy = nullptr;
    const int32_t* round = nullptr;
    const int32_t* count = nullptr;
};
struct ObsListOut {   // where launch_page_list leaves the same list (id == nullptr: none wanted)
    int32_t* id = nullptr;
    float *zx = nullptr, *zy = nullptr;
    int32_t *round = nullptr, *count = nullptr;
};
// Where a page lies.  The pages of a session on joint pages are one array of [5][32] floats; the pages of a SPLIT session on
// pages hold the two planes of MEANS only ([2][32] floats, 256 bytes) and lie in the session's two mean buffers, which are
// not neighbours in the store: pages below `half_pages` in the first, the others `gap` floats further on.
struct PageGeom {
    int planes = 5;                          // planes per page (5: means + covariances; 2: means)
    int64_t half_pages = (int64_t)1 << 62;   // pages at or beyond this index lie `gap` floats further on
    int64_t gap = 0;
};
struct PagedEkfArgs {
    PageGeom geom;
    // split session on pages (geom.planes == 2): the covariances come per class, as in EkfArgs
    const float* cov = nullptr;
    const float* covx = nullptr;
    int plane_stride = 0;             // floats between the planes of a class row (Lp)
    const int32_t* cls_in = nullptr;
    int32_t* cls_out = nullptr;
    uint32_t* cstamp = nullptr;
    uint32_t cstamp_now = 0;
    float* pool;             // [npages][planes][32]
    const int32_t* pt_in;    // [rows][nb] page tables of the ancestors
    int32_t* pt_out;         // [n][nb]    page tables of this frame's particles
    int nb;                  // pages per particle
    const int32_t* anc;      // source table of particle i (nullptr: i)
    int n, nlandmarks;
    const float *x, *y, *th;
    const float *obs_zx, *obs_zy;   // table form, NaN = not observed
    float meas_var;
    float* loglik;
    float* loglik_user;
    const int32_t *tpage, *tindex, *count;   // launch_page_list of the same table
    const int32_t* tmask;                    // ... bit s of tmask[t]: landmark s of touched page t is observed
    const int32_t* tbase;                    // ... tbase[t]: observations in the touched pages before t (tbase[T]: all)
    ObsListView ol;                          // the same observations as a list (the list form of the update)
    const int32_t* freelist;                 // particle i takes entries pool_state[base] + [i * T, (i + 1) * T)
    const int32_t* pool_state;               // written by launch_page_list of this frame
    uint32_t* stamp;                         // [npages]: every page a new table names gets stamp_now
    uint32_t stamp_now;
};
// pool_state (pool_state_words() int32 on the device): the free list's bookkeeping — how long it is, how much of it has
// been handed out, whether launch_free_list (to be called behind this launcher every frame) has to make a new one first
int pool_state_words();
// h_obs + votes (optional): the launch also takes SLAM_MAP_AUTO's sample — votes[2] (device): samples in a row with at most
// two sevenths / more than three eighths of the landmarks observed; h_obs (mapped host memory): {observed, L, seq, votes[0], votes[1]}
hipError_t launch_page_list(hipStream_t stream, const float* zx, const float* zy, int L, int nb, int32_t* tpage, int32_t* tindex,
                            int32_t* tmask, int32_t* tbase, int32_t* count, int n, int32_t* pool_state, int32_t* h_obs = nullptr,
                            uint32_t seq = 0, int32_t* votes = nullptr, int32_t* h_touched = nullptr,
                            const ObsListOut& ol = ObsListOut());
// the same sample as a launch of its own (one small workgroup), for sessions on rows
hipError_t launch_obs_count(hipStream_t stream, const float* zx, const float* zy, int L, int32_t* h_obs, uint32_t seq, int32_t* votes);
// form 0: every lane of a touched page runs the update arithmetic, one page after the other; form 1 (needs a.ol): the
// touched pages go through an LDS image, several at a time with all their loads in flight together, and the arithmetic
// runs once per observation, one lane each — the same bits.  touched_hint: pages to stage per pass (<= 0: a default)
hipError_t launch_ekf_paged(hipStream_t stream, const PagedEkfArgs& a, const EventPair* ev = nullptr, int form = 1,
                            int touched_hint = 0);
hipError_t launch_page_table_gather(hipStream_t stream, const int32_t* pt_in, int32_t* pt_out, int nb, const int32_t* anc, int n,
                                    uint32_t* stamp, uint32_t stamp_now);
// free list = pages whose stamp differs from `live`, the stamp of the last update (in no particular order); does
// nothing unless launch_page_list asked for it
// h_short (optional, mapped host memory): set to 1 when a list made anew turns out shorter than the reservation
hipError_t launch_free_list(hipStream_t stream, const uint32_t* stamp, int npages, uint32_t live, int32_t* freelist,
                            int32_t* pool_state, int32_t* h_short = nullptr);
// out[k] = anc[sel[k]] (anc == nullptr: sel[k])
hipError_t launch_compose_index(hipStream_t stream, const int32_t* sel, const int32_t* anc, int count, int32_t* out);
// rows -> pages page_base + j * nb + b behind identity tables; every other page of the pool goes on the free list
hipError_t launch_pages_from_rows(hipStream_t stream, const float* rows, int64_t row_stride, int plane_stride, int nlandmarks,
                                  int nb, int n, float* pool, int32_t* pt, int32_t* freelist, int npages, int32_t* pool_state,
                                  int page_base = 0, const PageGeom& geom = PageGeom());
hipError_t launch_rows_from_pages(hipStream_t stream, const float* pool, const int32_t* pt, int nb, const int32_t* anc, int n,
                                  float* rows, int64_t row_stride, int plane_stride, int nlandmarks, const PageGeom& geom = PageGeom());
hipError_t launch_pages_reset(hipStream_t stream, float* pool, int32_t* pt, int64_t nentries, int32_t* freelist, int npages,
                              int32_t* pool_state, const PageGeom& geom = PageGeom());
// split pages -> dense rows of 5 planes (means from the particle's pages of two planes, covariances from its class's rows)
hipError_t launch_rows_from_split_pages(hipStream_t stream, const float* pool, const PageGeom& geom, const int32_t* pt, int nb,
                                        const float* cov, const int32_t* cls, int Lp, const int32_t* idx, int count, float* rows,
                                        int64_t row_stride, int plane_stride, int nlandmarks);
// Sharded sessions: `want` pages for the rows about to be unpacked (pool_state then says where in the list they start and
// whether launch_free_list, to be called behind it, has to make a new list first), and the unpack itself: record p of
// `in` (pose + 5 x nlandmarks floats, as launch_migrate_pack writes them) -> table row n + p on fresh pages, stamped `live`
hipError_t launch_pool_reserve(hipStream_t stream, int32_t* pool_state, int64_t want);
// ... for a session on SPLIT PAGES: record p -> its means on fresh pages of two planes behind table row n + p, its covariances (and
// their determinant terms, meas_var) as class cls_free[cls_first + p] of its own, appended to the list of classes in use
hipError_t launch_migrate_unpack_split_pages(hipStream_t stream, const float* in, int total, int n, float* pose, int64_t pose_ld,
                                             float* pool, const PageGeom& geom, int32_t* pt, int nb, int nlandmarks,
                                             const int32_t* freelist, const int32_t* pool_state, uint32_t* stamp, uint32_t live,
                                             float* cov, float* covx, int32_t* cls, int Lp, float meas_var, const int32_t* cls_free,
                                             int cls_first, uint32_t* cstamp, uint32_t cstamp_now, int32_t* live_list, int32_t* live_cnt);
hipError_t launch_migrate_unpack_paged(hipStream_t stream, const float* in, int total, int n, float* pose, int64_t pose_ld,
                                       float* pool, int32_t* pt, int nb, int nlandmarks, const int32_t* freelist,
                                       const int32_t* pool_state, uint32_t* stamp, uint32_t live);

hipError_t launch_logweight(hipStream_t stream, const float* score, const float* loglik, float gain, int n,
                            float* logw, float* block_max_scratch, float* d_max, const float* carry = nullptr,
                            const int32_t* prev_resampled = nullptr, const CovArgs* cov = nullptr, int cov_bound = 0);
// (cov: the same launch carries launch_cov_update(cov, cov_bound) in workgroups of its own — cov_update_body.h)
int logweight_scratch_elems(int n);
int logweight_scratch_floats();   // size of block_max_scratch: block maxima + two words, zero-initialised once
hipError_t launch_quantise_weights(hipStream_t stream, const float* logw, const float* d_max, int n, uint64_t* wq,
                                   uint64_t* d_sum);

// The resample gate (ESS-gated resampling; oracle: orc_ess_resample).  frac_q16 = threshold * 65536, 0 = no gate
// (resample every frame).  Where the verdict goes: a device flag and, optionally, mapped host memory
// {int32 resampled, uint32 sequence number}.
// Where a resample stage reports roughly how many distinct ancestors it left: counter = one 8-byte-aligned pair of words
// (device, zeroed once, left zeroed), h_out = {count, n} in mapped host memory.
struct HeadsOut {
    unsigned int* counter = nullptr;
    int32_t* h_out = nullptr;
};
// The gate's small device buffer, int32 words: [0] "the last resample stage did resample" (the flag logweight_kernel reads),
// [1] pad, [2] the ticket of quantise_scan_kernel's shard sums, [3] pad, [4..9] its three 64-bit accumulators (8-byte aligned)
enum { kGateFlagWord = 0, kGateTicketWord = 2, kGateBufWords = 16 };
struct GateOut {
    int32_t* d_flag = nullptr;
    int32_t* h_flag = nullptr;
    uint32_t seq = 0;
};
// fused frame-loop form (quantise + tile scan; offsets straight from the tile-local scan).  The scan state is
// cdf_local[n] followed by tile_total[ntiles] | tile_s16[ntiles] | tile_q16[ntiles] (the last two only with a gate:
// carry != nullptr).  With a gate d_sum receives three values (total, S, Q) and d_shard_totals holds such triples.
hipError_t launch_quantise_scan(hipStream_t stream, const float* logw, const float* d_max, const float* block_max,
                                int nblock_max, int n, uint64_t* cdf_local, uint64_t* tile_total, uint64_t* d_sum,
                                float* carry = nullptr, uint64_t* tile_s16 = nullptr, uint64_t* tile_q16 = nullptr,
                                unsigned int* ticket = nullptr);
hipError_t launch_offspring_from_scan(hipStream_t stream, const uint64_t* cdf_local, const uint64_t* tile_total, int n,
                                      const uint64_t* d_base, const uint64_t* d_total, const uint64_t* d_shard_totals,
                                      int rank, int world, uint64_t seed, uint32_t frame, int64_t n_total,
                                      int32_t* first, uint32_t frac_q16 = 0, const GateOut& gate = GateOut());
hipError_t launch_prefix_sum(hipStream_t stream, const uint64_t* in, int n, uint64_t* out, uint64_t* block_scratch);
int prefix_sum_scratch_elems(int n);
int scan_tile_count(int n);   // 2048-element tiles of the fused quantise + scan
hipError_t launch_offspring_offsets(hipStream_t stream, const uint64_t* cdf, int n, const uint64_t* d_base,
                                    const uint64_t* d_total, uint64_t seed, uint32_t frame, int64_t n_total,
                                    int32_t* first);
hipError_t launch_ancestors(hipStream_t stream, const int32_t* first_all, int64_t n_total, int64_t slot0, int nslots,
                            int32_t* anc);
// single GPU: offspring offsets + ancestors in one launch (n up to 8M; beyond that use the two launches above)
bool ancestors_from_scan_fits(int n);
hipError_t launch_ancestors_from_scan(hipStream_t stream, const uint64_t* cdf_local, const uint64_t* tile_total, int n,
                                      uint64_t seed, uint32_t frame, int32_t* anc, uint32_t frac_q16 = 0,
                                      const GateOut& gate = GateOut(), const HeadsOut& heads = HeadsOut());
// multi-GPU resample (see pf_kernels.hip): per-peer slot runs, offsets in the packed exchange buffers
enum { kMaxRanks = 16 };
struct MigratePlan {
    int64_t lo[kMaxRanks];       // pack: prefix count at the first particle sent to peer q (send_base); unpack: unused
    int32_t off[kMaxRanks + 1];  // running particle offset of peer q's run in the buffer (off[world] = total)
    int32_t world;
};
int shard_scan_words(int n);   // int32 words of scratch the two launchers below share
// host_plan / host_flag / seq (optional): also deliver the plan to mapped pinned host memory and release `seq`
hipError_t launch_ancestors_sharded(hipStream_t stream, const int32_t* first_all, int64_t n_total, int n, int rank,
                                    int world, int32_t* scratch, int32_t* plan, int32_t* src, int32_t* pose_idx,
                                    int32_t* host_plan, uint32_t* host_flag, uint32_t seq, int recv_cap,
                                    int32_t* host_heads = nullptr);
hipError_t launch_migrate_pack(hipStream_t stream, const int32_t* scratch, int n, const MigratePlan& plan,
                               const float* pose, int64_t pose_ld, const float* map, int64_t row_stride,
                               int plane_stride, int nlandmarks, float* out,
                               const int32_t* pt = nullptr, int nb = 0,
                               const float* split_cov = nullptr, const int32_t* split_cls = nullptr,
                               const PageGeom& geom = PageGeom());   // pt AND split_cls: split pages (means behind pt in pages of geom)
hipError_t launch_migrate_unpack(hipStream_t stream, const float* in, const MigratePlan& plan, int n, float* pose,
                                 int64_t pose_ld, float* map, int64_t row_stride, int plane_stride, int nlandmarks);
hipError_t launch_argmax(hipStream_t stream, const float* v, int n, int32_t* idx_out, float* val_out);
// heaviest particle {logw, global id (int bits), x, y, theta} -> out5 (device) and optionally mapped host memory + seq
hipError_t launch_best_particle(hipStream_t stream, const float* v, int n, const float* px, const float* py,
                                const float* pth, int64_t first_id, float* out5, float* h_out5, uint32_t* h_seq,
                                uint32_t seq);
// exact fixed-point sums of the population {x, y: 2^-32; sin, cos of (theta - ref): 2^-30} -> out4 (device), optionally
// mapped host memory + seq; acc[4] / ticket: zero-initialised device scratch the kernel leaves zeroed
hipError_t launch_pose_sums(hipStream_t stream, const float* x, const float* y, const float* th, const int32_t* idx,
                            int n, float ref_th, unsigned long long* acc, unsigned int* ticket, long long* out4,
                            long long* h_out4, uint32_t* h_seq, uint32_t seq);
// measurement support: rows of 5 x plane_stride floats copied with the update's access shape (slam_profile_copy_ceiling)
hipError_t launch_copy_rows(hipStream_t stream, const float* in, float* out, int n, int plane_stride);
hipError_t launch_gather_f32(hipStream_t stream, const float* src, const int32_t* idx, int n, float* dst);
hipError_t launch_gather_map(hipStream_t stream, const float* in, float* out, int64_t in_row_stride,
                             int64_t out_row_stride, int in_plane_stride, int out_plane_stride, int nlandmarks,
                             const int32_t* idx, int n);

}  // namespace slam
