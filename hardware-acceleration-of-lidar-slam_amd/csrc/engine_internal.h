// engine_internal.h — the engine object shared by the translation units that implement the C ABI
// (engine.hip, mapper.hip).  Not part of the public interface.
#pragma once

#include <hip/hip_runtime.h>

#include <vector>

#include "kernels.h"

using slam::EventPair;

struct slam_comm;

namespace slam_detail {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 256;
        hipError_t err = hipMalloc(&p, want);
        if (err == hipSuccess) cap = want;
        return err;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

struct GridSlot {
    bool ready = false;
    slam_grid_meta meta{};
    const float* d_edt = nullptr;   // owned (edt_buf) or adopted
    DevBuf occ_buf, edt_buf;
    // the byte-per-cell copy the many-pose scorers gather from (kernels.h: ScoreGrid::packed), made on the first such call
    // after the grid changed: 0 = not made yet, 1 = in use, 2 = this grid has no such copy (values the table cannot give back)
    int packed_state = 0;
    DevBuf packed_buf, table_buf;   // table_buf: 256 floats + the two flag words of launch_edt_pack
};

constexpr int kLattice = 27;
// device/host staging layout of one FastMatch call (floats):
//   in : X[27] Y[27] CT[27] ST[27] LAST[4]
//   out: SCORE[27] COUNT[27] NLAST[1] HITS[SLAM_MAX_BEAMS]
constexpr int kFmIn = 4 * kLattice + 4;
constexpr int kFmOut = 2 * kLattice + 1 + SLAM_MAX_BEAMS;
// the chained pair of FastMatch calls (slam_engine_fastmatch_pair), behind the arrival flag's 4 words: in — the nine headings of
// the second call's lattices as cos[9] | sin[9], its step, a spare word; out — the first call's scores and counts
constexpr int kFmPairIn = 20;
constexpr int kFmPair = kFmPairIn + 2 * kLattice;
constexpr unsigned kStageSlots = 8;
constexpr size_t kStageFloats = 3 * SLAM_MAX_OBS > 2 * SLAM_MAX_BEAMS ? 3 * SLAM_MAX_OBS : 2 * SLAM_MAX_BEAMS;

}  // namespace slam_detail

using namespace slam_detail;

struct slam_engine {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    char err[512] = { 0 };

    GridSlot grid[SLAM_MAX_GRID_SLOTS];

    const float *d_bx = nullptr, *d_by = nullptr;   // owned (scan_buf) or adopted
    int nbeams = -1;
    DevBuf scan_buf;

    // observations of the current frame as a table indexed by landmark (zx NaN = not observed)
    DevBuf obs_buf;   // zx[L] zy[L] when uploaded from the host
    int obs_nlandmarks = -1;
    const float *d_obs_zx = nullptr, *d_obs_zy = nullptr;

    DevBuf fm_buf;             // kFmIn + kFmOut floats
    DevBuf fm_work;            // 27 x SLAM_MAX_BEAMS floats: per-candidate hit rows of the lattice kernel
    float* h_fm = nullptr;     // pinned + mapped: lattice candidates in, results out, then one uint32 arrival flag
    float* d_hfm = nullptr;    // the same memory as the device sees it (zero-copy FastMatch I/O)
    uint32_t fm_seq = 0;
    int32_t* h_plan = nullptr;   // pinned + mapped: exchange plan of the last slam_ancestors_sharded_dev, then one arrival flag
    int32_t* d_hplan = nullptr;
    uint32_t plan_seq = 0;
    int plan_world = 0;
    int exch_cap = 0x7fffffff;   // staging rows a rank can take in one exchange (slam_exchange_set_capacity)
    DevBuf scratch;            // per-call temporaries of the *_dev stages
    DevBuf bmax_buf;           // block maxima left by slam_logweight_dev (read by slam_quantise_scan_dev)
    int bmax_count = 0, bmax_n = -1;
    DevBuf scan_state;         // tile-local CDF u64[n] + tile totals, left by slam_quantise_scan_dev
    int scan_n = -1;
    DevBuf shard_buf;          // flag scans of the last slam_ancestors_sharded_dev call (read by slam_migrate_pack_dev)
    int shard_n = -1;
    DevBuf first_buf;          // scratch `first` array of slam_ancestors_from_scan_dev's large-n fallback
    DevBuf ll_buf;             // log-likelihood [n] of the last EKF call
    int ll_n = -1;
    // resample gate (slam_resample_gate_set): threshold, the device flag "the last resample stage did resample", the
    // same verdict in mapped host memory {int32 resampled, uint32 sequence}, and the weights carried over a frame
    // that did not resample
    uint32_t gate_frac_q16 = 0;
    DevBuf gate_buf;           // int32 flag
    int32_t* h_gate = nullptr;
    int32_t* d_hgate = nullptr;
    uint32_t gate_seq = 0;
    DevBuf carry_buf;          // float[n]
    int carry_n = -1;

    // feedback of the resample stages: roughly how many distinct ancestors the last one left {count, n} (mapped host
    // memory, read without synchronisation: it only steers the choice between two equivalent EKF kernels)
    int32_t* h_heads = nullptr;
    int32_t* d_hheads = nullptr;
    DevBuf heads_buf;          // {count, ticket}

    slam::HeadsOut heads_out()
    {
        slam::HeadsOut h;
        h.counter = heads_buf.as<unsigned int>();
        h.h_out = d_hheads;
        return h;
    }
    // in-place updates (frames that keep their population) have two equivalent kernels as well: whole rows in batches of
    // 128 landmarks, or only the observed landmarks from a compact list (obs_list: ids | zx | zy | rounds, [L] each, then
    // {nobs, highest round}).  The list is built at most once per observation table; {nobs, L} of the last build sit in
    // mapped host memory and steer the choice for the following frames (read without synchronisation).
    DevBuf obs_list;
    bool obs_list_valid = false;
    bool obs_table_owned = false;   // the table is the engine's own copy (slam_obs_upload_host), not the caller's arrays
    int32_t* h_obs = nullptr;
    int32_t* d_hobs = nullptr;
    // 128 bytes of mapped host memory the engine's (one) particle-filter session delivers its results through.  It belongs to
    // the ENGINE, not to the session: freeing pinned host memory makes the driver hold the process's queues for 65-80 ms some
    // 10-50 ms later (profiles/r03_stall_trigger.txt) — a session that came and went would stall the frames of the next one.
    void* h_block = nullptr;        // the one pinned allocation every h_* pointer of the engine points into
    void* h_pf_res = nullptr;
    void* d_hpf_res = nullptr;
    bool frame_fusion = true;       // slam_frame_fusion_set
    int64_t front_launches = 0;
    int32_t front_last[2] = { 0, 0 };   // slam_frame_front_last: particles per updating wavefront, lanes per pose
    int ekf_inplace_form = -1;   // slam_ekf_inplace_form_set: -1 by the feedback, 0 whole rows, 1 observed landmarks only
    int64_t ekf_inplace_launches[2] = { 0, 0 };
    slam_comm* comm = nullptr; // the communicator made on this engine, if any: host-side waits poll it for failures
    int live_sessions = 0;     // slam_pf sessions alive on this engine (at most one: the stages keep per-population state here)
    bool pf_paged = false;     // slam_pf_paged_set: sessions made from now on keep their maps as copy-on-write pages
    int ekf_form = -1;         // slam_ekf_form_set: -1 choose by the feedback, 0 row per wavefront, 1 / 2 grouped by 4 / 2
    int64_t ekf_form_launches[2] = { 0, 0 };   // out-of-place launches so far: [0] one wavefront per particle, [1] grouped
    // particles per wavefront of an out-of-place update that gathers through `anc` (0 = one wavefront per particle).  With the
    // covariance part of the update hoisted (ekf_prepare) the kernel is memory-bound and behaves like the grouped pure copy of
    // profiles/copy_ceiling.hip: 2 rows per wavefront beat 4 and 8 (update alone at 64k x 500: 133 | 145 | 155 us; fused
    // front at 512k x 5000: 8.10 against 9.00 ms, at 64k x 2000: 0.558 against 0.578 ms per frame) — except in the fused
    // front of a small frame, where 4 keep the number of updating workgroups per scoring workgroup low (64k x 500: 143
    // against 160 us; 3 particles per wavefront: 143.5 against 142.6 us fused, 138 against 134 us alone) as long as neighbours
    // share ancestors (fewer than 3 distinct in 10 slots, as far as the last resample stage reported).
    // Split layout: the kernel is bound by its vector instructions and by what a wavefront does once per pass whatever its
    // group (observation flags, the class's covariances, its own start-up with the motion sample), so larger groups win:
    // fused front at 64k x 500: 2 particles per wavefront 116 us, 4: 99 us, 8: 92.6 us.
    int ekf_group_size(int n, bool has_anc, int plane_stride, bool fused, bool split = false) const
    {
        if (ekf_form >= 0) return ekf_form == 0 ? (split ? 2 : 0) : (ekf_form == 2 ? 2 : 4);
        if (split) {
            const int heads = h_heads[0], hn = h_heads[1];
            return hn == n && (int64_t)heads * 10 < (int64_t)n * 3 ? 8 : 4;
        }
        if (!has_anc) return 0;
        if (!fused) return 2;
        const int heads = h_heads[0], hn = h_heads[1];
        const bool few_distinct = hn == n && (int64_t)heads * 10 < (int64_t)n * 3;
        return few_distinct && (int64_t)n * plane_stride < (int64_t)1 << 26 ? 4 : 2;
    }

    slam::GateOut gate_next()
    {
        slam::GateOut g;
        g.d_flag = gate_buf.as<int32_t>();
        g.h_flag = d_hgate;
        g.seq = ++gate_seq;
        return g;
    }
    // pinned staging ring for the per-frame sensor uploads: one host-to-device copy per upload, and the
    // host only waits if kStageSlots uploads are still in flight
    float* h_stage = nullptr;
    hipEvent_t stage_ev[8] = {};
    unsigned stage_next = 0;

    float* stage_acquire()
    {
        const unsigned k = stage_next++ % kStageSlots;
        if (stage_ev[k]) (void)hipEventSynchronize(stage_ev[k]);
        return h_stage + (size_t)k * kStageFloats;
    }
    hipError_t stage_release(const float* slot)
    {
        const unsigned k = (unsigned)((slot - h_stage) / kStageFloats);
        if (!stage_ev[k]) {
            hipError_t err = hipEventCreateWithFlags(&stage_ev[k], hipEventDisableTiming);
            if (err != hipSuccess) return err;
        }
        return hipEventRecord(stage_ev[k], stream);
    }
    DevBuf host_io[6];         // temporaries of the *_host convenience calls

    // per-kernel HIP-event timing (slam_profile_*)
    int prof_mask = 0;
    std::vector<EventPair> prof_pool[SLAM_PROF_COUNT];   // grown on demand, reused after each read
    size_t prof_used[SLAM_PROF_COUNT] = {};
    EventPair prof_cur{};

    const EventPair* prof_next(int k)
    {
        if (!(prof_mask & (1 << k))) return nullptr;
        auto& pool = prof_pool[k];
        if (prof_used[k] == pool.size()) {
            EventPair p;
            if (hipEventCreate(&p.start) != hipSuccess || hipEventCreate(&p.stop) != hipSuccess) return nullptr;
            pool.push_back(p);
        }
        return &pool[prof_used[k]++];
    }
};


// HIP events around whatever is enqueued on the engine's stream during the scope's lifetime (slam_profile_*)
struct ProfScope {
    slam_engine* e;
    const EventPair* ev;
    ProfScope(slam_engine* e_, int kernel) : e(e_), ev(e_->prof_next(kernel))
    {
        if (ev) (void)hipEventRecord(ev->start, e->stream);
    }
    ~ProfScope()
    {
        if (ev) (void)hipEventRecord(ev->stop, e->stream);
    }
    ProfScope(const ProfScope&) = delete;
    ProfScope& operator=(const ProfScope&) = delete;
};

// helpers implemented in engine.hip
// slam_logweight_dev / slam_logweight_ekf_dev (use_ekf) with, optionally, a split session's covariance classes brought up to
// date by workgroups of the same launch (launch_logweight)
extern "C" int slam_logweight_cov_dev(slam_engine* e, const float* d_score, bool use_ekf, float score_gain, int n, float* d_logw, float* d_max,
                           const slam::CovArgs* cov, int cov_bound);
// slam_motion_score_dev with a rider (defined in engine.hip next to it)
extern "C" int slam_motion_score_rider_dev(slam_engine* e, int slot, const float* d_src_x, const float* d_src_y, const float* d_src_th,
                                           const int32_t* d_anc, float* d_x, float* d_y, float* d_th, int n, int64_t first_id,
                                           const float dp[3], const float sigma[3], uint64_t seed, uint32_t frame, float* d_score,
                                           int32_t* d_count, const slam::FreeListRider* rider, bool* rode);
// slam_motion_score_dev + slam_ekf_update_dev (out of place, through the resample indices d_anc) as ONE launch
// (launch_frame_front).  *launched = false: the shapes do not fit or fusion is off — nothing was issued, the caller makes
// the two calls.  Used by the slam_pf session for single-GPU frames on rows.
extern "C" int slam_frame_front_dev(slam_engine* e, int slot, const float* d_src_x, const float* d_src_y, const float* d_src_th,
                         const int32_t* d_anc, float* d_x, float* d_y, float* d_th, int n, int64_t first_id, const float dp[3],
                         const float sigma[3], uint64_t seed, uint32_t frame, float* d_score, int32_t* d_count,
                         const float* d_map_in, float* d_map_out, int64_t row_stride, int plane_stride, int nlandmarks,
                         float meas_var, bool* launched, const slam::SplitIO* split = nullptr);
// the out-of-place landmark update on the SPLIT layout (kernels.h: EkfArgs): d_mean_in / d_mean_out are rows of two planes
// (row_stride >= 2 * plane_stride), the covariances come per class through `split`; otherwise slam_ekf_update_dev
extern "C" int slam_ekf_split_dev(slam_engine* e, const float* d_mean_in, float* d_mean_out, int64_t row_stride, int plane_stride,
                       int nlandmarks, const float* d_x, const float* d_y, const float* d_th, const int32_t* d_anc, int n,
                       float meas_var, const slam::SplitIO* split);
int slam_engine_fail_hip(slam_engine* e, hipError_t err, const char* what);
slam::ScoreGrid slam_engine_score_grid(const slam_engine* e, int slot);
// FastMatch on grid `slot` with the beam count read from device memory (d_nbeams, at most nbeams_max) and the
// scan at d_bx/d_by; optionally mirrors the hit scratch into d_hits_persist.  Synchronises.
int slam_engine_fastmatch(slam_engine* e, int slot, const float* d_bx, const float* d_by, int nbeams_max,
                          const int32_t* d_nbeams, const float pose[3], const float res[3], float out_pose[3],
                          float* best_hits, int32_t* best_hits_size, float* best_score, float* d_hits_persist);
// ... and FastMatch(pose, res1) on slot1 followed by FastMatch2(its result, res2) on slot2 as ONE round trip (engine.hip)
int slam_engine_fastmatch_pair(slam_engine* e, int slot1, int slot2, const float* d_bx, const float* d_by, int nbeams_max,
                               const int32_t* d_nbeams, const float pose[3], const float res1[3], const float res2[3],
                               float out_pose[3], int32_t* best_hits_size, float* d_hits_persist);

// slam_migrate_pack_dev for a session with paged maps (d_pt: page tables of nb entries, d_map: the page pool)
extern "C" int slam_migrate_pack_paged(slam_engine* e, int n_local, int rank, int world, const int32_t* plan, const float* d_pose,
                            int64_t pose_ld, const float* d_map, int64_t row_stride, int plane_stride, int nlandmarks,
                            float* d_out, const int32_t* d_pt, int nb, const float* d_split_cov = nullptr,
                            const int32_t* d_split_cls = nullptr,   // split layout: d_map = the means (row_stride >= 2 planes)
                            const slam::PageGeom* geom = nullptr);   // split pages (d_pt and d_split_cls): d_map = the pool of mean pages

#define SLAM_HIP_TRY(e, call)                                                     \
    do {                                                                          \
        hipError_t err__ = (call);                                                \
        if (err__ != hipSuccess) return slam_engine_fail_hip((e), err__, #call); \
    } while (0)
