/*
 * slam_hip.h — C ABI of the MI355X (gfx950) scan-matching / particle-filter engine.
 *
 * This is the drop-in boundary for the hot path of circuitpotato/Hardware-Acceleration-of-LIDAR-SLAM:
 * the functions below replace, one stage per call, what Subsystem_1/main_accelerated.c does in
 *   euclidean_distance_transform / euclidean_distance_transform2   (main_accelerated.c:215-283)
 *   FastMatch / FastMatch2                                        (main_accelerated.c:396-824)
 * and add the particle-filter stages the north star asks for around them (motion sample,
 * per-particle x per-landmark 2x2 EKF, weight normalisation, systematic resample), for which the
 * reference has no code.  INTEGRATION.md shows the edits a maintainer of the
 * reference makes to call these from main_accelerated.c.
 *
 * Conventions
 *  - plain C, no C++/torch types; every function returns a slam_status (0 = OK, < 0 = error) and
 *    never aborts or exits (the reference's functions are all `void` and print-and-continue,
 *    main.c:15-20; we return codes instead).
 *  - `slam_engine` is an opaque handle created once and passed to every call — the same shape as the
 *    `accel` handle the reference's FPGA variant threads through OccupationalGrid/EDT
 *    (Submodule_2/Hadrware_acclereated.cpp:236,260,284,842-845).  One engine per host thread / GPU;
 *    calls on one engine are serialised on one HIP stream.
 *  - `*_host` entry points take caller-owned HOST buffers (as the reference's functions do), copy,
 *    launch, copy back and synchronise.  `*_dev` entry points take DEVICE pointers, are asynchronous
 *    on the engine's stream and keep everything resident in HBM; they are what a frame loop with
 *    many particles uses.  The boundary is crossed once per frame stage, never per element (the
 *    reference's FPGA path crossed it once per distance, Hadrware_acclereated.cpp:223-233).
 *  - there is NO CPU fallback: if no gfx950 device is usable the calls fail with SLAM_ERR_NO_DEVICE.
 *  - float arithmetic of the reference-pinned stages (EDT, score) is bit-exact with the reference:
 *    same operation order, no FMA contraction.
 */
#ifndef SLAM_HIP_H
#define SLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAM_ABI_VERSION 5

typedef enum {
    SLAM_OK = 0,
    SLAM_ERR_NO_DEVICE = -1,   /* no usable gfx950 GPU / HIP runtime error at init */
    SLAM_ERR_INVALID_ARG = -2, /* null pointer, negative size, slot out of range, ... */
    SLAM_ERR_HIP = -3,         /* a HIP runtime call failed; slam_last_error() has the text */
    SLAM_ERR_NOT_READY = -4,   /* stage called before its inputs were provided (e.g. no grid in slot) */
    SLAM_ERR_CAPACITY = -5,    /* request exceeds a fixed capacity (beams, lattice size, ...) */
    SLAM_ERR_COMM = -6         /* an exchange between ranks failed (RCCL error, or a rank of an in-process group
                                  did not arrive); slam_last_error() has the text */
} slam_status;

typedef struct slam_engine slam_engine;

/* Bookkeeping of one occupancy/EDT grid: the fields the reference keeps beside each grid in
 * `MyGrid` (main.c:200-213): grid_size{rows,cols}, pixel_size, top_left_corner; `ld` is the leading
 * dimension of the row-major storage (fixed 200 / 400 in the reference). */
typedef struct {
    int32_t rows, cols, ld;
    float pixel;
    float min_x, min_y;
} slam_grid_meta;

/* ------------------------------------------------------------------ engine lifetime */

int slam_abi_version(void);
const char *slam_status_string(int status);
/* text of the most recent HIP error seen by this engine (empty string if none); e == NULL: why the calling thread's last
 * slam_engine_create failed */
const char *slam_last_error(const slam_engine *e);

/* device: HIP device ordinal.  Fails with SLAM_ERR_NO_DEVICE when there is none.
 * A host that times its first frames should know: the engine makes ONE allocation of pinned host memory here (every mapped
 * result buffer and the upload staging ring are carved out of it), and the driver answers such an allocation some 10-50 ms
 * later by holding the process's queues for 65-80 ms (profiles/r03_stall_trigger.txt).  Results are
 * unaffected; 50 ms between slam_engine_create and the first timed frame keep the hold out of the timing (bench.py sleeps
 * 250 ms).  Creating and destroying sessions on a live engine allocates no pinned memory. */
int slam_engine_create(int device, slam_engine **out);
int slam_engine_destroy(slam_engine *e);
/* Run the engine on a caller-provided hipStream_t (e.g. the framework's current stream) instead of
 * its own.  NULL means what it means to HIP: the default (null) stream.  SLAM_OWN_STREAM restores the
 * engine's own non-blocking stream (the initial state). */
#define SLAM_OWN_STREAM ((void *)(intptr_t)-1)
int slam_engine_set_stream(slam_engine *e, void *hip_stream);
int slam_engine_sync(slam_engine *e);

/* Per-kernel timing with HIP events recorded on the engine's stream around the named kernel's
 * launches only (not around host work or neighbouring kernels).  Off by default.  `mask` selects the
 * kernels: bit k = slam_prof_kernel k (an event pair costs a few microseconds of stream time per launch,
 * so time only what you need); 0 switches timing off.  slam_profile_read
 * synchronises the stream, returns the summed duration and the launch count since the last reset
 * and resets the counters.  (The reference's only timer is clock() around the whole run,
 * main.c:826-827, 971-973.) */
typedef enum {
    SLAM_PROF_SCORE = 0,        /* score_poses_kernel (with or without the motion sample), lattice kernels */
    SLAM_PROF_EDT = 1,          /* edt kernels */
    SLAM_PROF_EKF = 2,          /* the landmark update of a frame, whichever kernel runs it (rows, grouped, list, pages) */
    SLAM_PROF_WEIGHTS = 3,      /* log-weights + block maxima (+ the maximum's finalize launch on several GPUs); a split session's
                                   covariance classes are brought up to date by workgroups of the same launch */
    SLAM_PROF_SCAN = 4,         /* quantise + prefix sum (+ ESS sums) */
    SLAM_PROF_ANCESTORS = 5,    /* offspring offsets / ancestor search (single GPU: one launch) */
    SLAM_PROF_PLAN = 6,         /* several GPUs: global ancestor search, flag scans, exchange plan, local gather index */
    SLAM_PROF_PACK = 7,         /* several GPUs: migrating rows into the send buffer */
    SLAM_PROF_UNPACK = 8,       /* several GPUs: received rows into the staging tail */
    SLAM_PROF_COLLECTIVES = 9,  /* every exchange between ranks (all-reduce, all-gathers, send/recv), as the stream sees them */
    SLAM_PROF_PAGES = 10,       /* map bookkeeping: paged maps' touched-page list, table gathers, the free list where it is a launch of
                                   its own (one GPU: it travels in the scorer's launch); split maps' gathers on frames without observations */
    SLAM_PROF_EKF_TAIL = 11,    /* several GPUs, split maps: the part of the landmark update that waits for the exchange (the
                                   groups with an ancestor in the staging tail); the rest went out with the score (SLAM_PROF_EKF) */
    SLAM_PROF_COUNT = 12
} slam_prof_kernel;
int slam_profile_enable(slam_engine *e, int mask);
int slam_profile_read(slam_engine *e, int kernel, double *total_ms, int64_t *launches);
/* What an EMPTY start/stop bracket measures on this stream (mean of 64 back-to-back pairs, ms): the part of a
 * bracketed duration that is event bookkeeping rather than kernel time.  Synchronises. */
int slam_profile_bracket_overhead(slam_engine *e, double *overhead_ms);
/* What a PURE COPY with the landmark update's access shape reaches on this GPU: `rows` rows of five planes of
 * `plane_stride` floats (a multiple of 128) are copied from d_src to d_dst `reps` times (one wavefront per row, 256-byte
 * wave accesses, all loads of two batches before their stores, XCD-contiguous workgroup numbering, streaming stores —
 * the addressing of the update without its arithmetic); *ms_per_copy is the average duration from HIP events.  Any
 * kernel that reads 20 B and writes 20 B per (particle, landmark) is bounded by it: measurement support, not a stage. */
/* Self-check of the landmark update's reciprocal (csrc/ekf_math.h): the update computes 1 / det without the scaling and
 * fix-up steps of a general IEEE division whenever det's exponent lies in [-60, 60]; this walks every float of that range
 * on the device (2 x 121 x 2^23 values) and compares with the compiler's correctly rounded division.  *mismatches must
 * come back 0. */
int slam_selftest_reciprocal(slam_engine *e, int64_t *mismatches, int64_t *checked);
int slam_profile_copy_ceiling(slam_engine *e, const float *d_src, float *d_dst, int64_t rows, int plane_stride, int reps,
                              double *ms_per_copy);

/* ------------------------------------------------------------------ EDT */

/* Capped exact Euclidean distance transform of occ[0..rows)[0..cols) (row-major, leading dimension
 * ld, non-zero = occupied): out = 0 on occupied cells, else min(cap, sqrtf(min d^2)) in cells; cells
 * outside rows x cols are not written.  Replaces euclidean_distance_transform{,2}
 * (main.c:223-269, main_accelerated.c:215-283). */
int slam_edt_dev(slam_engine *e, const int32_t *d_occ, int ld, int rows, int cols, float cap, float *d_out);
int slam_edt_host(slam_engine *e, const int32_t *occ, int ld, int rows, int cols, float cap, float *out);

/* ------------------------------------------------------------------ engine-resident grids + scan
 * (what the reference keeps in the globals `occ_grid` and `scan`) */

enum { SLAM_MAX_GRID_SLOTS = 4, SLAM_MAX_BEAMS = 4096 };

/* Upload an occupancy grid, build its EDT on the device and keep it as grid `slot` (0 = the coarse
 * grid FastMatch reads, 1 = the fine grid FastMatch2 reads).  When edt_out != NULL the EDT is also
 * copied back in the same [.. ][ld] layout (only the rows x cols rectangle is written).
 * = the tail of OccupationalGrid (main.c:355-362). */
int slam_grid_upload_host(slam_engine *e, int slot, const int32_t *occ, const slam_grid_meta *meta, float cap,
                          float *edt_out);
/* Replace pixel / min_x / min_y of a grid already in `slot` (rows, cols, ld must be unchanged).  The
 * reference runs its EDTs before it records pixel_size and top_left_corner (main.c:355-362); an
 * adapter with the reference's EDT signature therefore learns them one step later. */
int slam_grid_set_meta(slam_engine *e, int slot, const slam_grid_meta *meta);
/* Adopt an EDT that is already on the device (not copied; caller keeps it alive).
 * The scorers of many poses (slam_score_poses_*, slam_motion_score_dev, slam_pf_step with >= 3 072 poses) gather from a packed
 * copy of the grid — one byte per cell, made on their first call after the grid changed (the capped EDT holds only 0,
 * sqrtf of small integers and the cap: main.c:223-269; a grid with other values keeps the float path) — so call
 * slam_grid_set_dev again after rewriting the cells of an adopted grid (slam_edt_dev into the adopted buffer is noticed
 * by itself).  Results are bit-identical either way. */
int slam_grid_set_dev(slam_engine *e, int slot, const float *d_edt, const slam_grid_meta *meta);
/* Sensor-frame cartesian beams of the current scan = scan.x / scan.y (main.c:60-69). */
int slam_scan_upload_host(slam_engine *e, const float *bx, const float *by, int nbeams);
int slam_scan_set_dev(slam_engine *e, const float *d_bx, const float *d_by, int nbeams);

/* ------------------------------------------------------------------ scan-match score (row A7) */

/* Score every pose against grid `slot` with the current scan:
 *   score[i] = in-order float sum of EDT[cell(pose_i (+) beam_b)] over in-bounds beams,
 *   count[i] = number of in-bounds beams                       (main.c:459-518, Appendix A.5).
 * Heading trig: the `_cs` form takes cos/sin per pose computed by the caller (the reference computes
 * them with libm on the host, main.c:433-435; this form is bit-exact with the reference for any
 * pose); the theta form computes them on the device with the engine's specified polynomial
 * (DESIGN.md "device trig"), bit-exact with the oracle's restatement of the same polynomial. */
int slam_score_poses_cs_dev(slam_engine *e, int slot, const float *d_x, const float *d_y, const float *d_ct,
                            const float *d_st, int nposes, float *d_score, int32_t *d_count);
int slam_score_poses_dev(slam_engine *e, int slot, const float *d_x, const float *d_y, const float *d_theta,
                         int nposes, float *d_score, int32_t *d_count);
int slam_score_poses_cs_host(slam_engine *e, int slot, const float *x, const float *y, const float *ct,
                             const float *st, int nposes, float *score, int32_t *count);
int slam_score_poses_host(slam_engine *e, int slot, const float *x, const float *y, const float *theta, int nposes,
                          float *score, int32_t *count);
/* In-bounds EDT values of ONE pose in beam order (what the reference leaves in
 * FastMatchParameters.bestHits, main.c:515); hits must hold nbeams floats. */
int slam_pose_hits_host(slam_engine *e, int slot, float x, float y, float ct, float st, float *hits, int32_t *count);

/* Drop-in for FastMatch (slot 0) / FastMatch2 (slot 1), main.c:381-596 / 598-809, quirks included:
 * 27-pose lattice laid out once around `pose` with step res[0] in x AND y and res[2] in theta
 * (res[1] is never read), strict '<' arg-min in theta-major, x, y-minor order; *best_hits_size =
 * in-bounds count of the BEST candidate while best_hits[] is left as the reference leaves its shared
 * scratch: every candidate overwrote the prefix [0, its count), last writer wins, entries beyond the
 * longest candidate keep the caller's previous content.  best_hits must hold
 * nbeams floats and should persist between calls like FastMatchParameters.bestHits; best_score may
 * be NULL. */
int slam_fastmatch_host(slam_engine *e, int slot, const float pose[3], const float res[3], float out_pose[3],
                        float *best_hits, int32_t *best_hits_size, float *best_score);

/* ------------------------------------------------------------------ particle-filter stages
 * (rows A9-A12: no counterpart in the reference; specified in DESIGN.md and oracle/slam_oracle_pf.c).
 * Particles are SoA float arrays x[], y[], theta[]; `first_id` is the global index of element 0 of
 * this shard (0 on a single GPU) so that results do not depend on how particles are sharded. */

/* A9: x,y,theta[i] = src(x,y,theta)[anc ? anc[i] : i] + dp + eps_i, eps ~ N(0, diag(sigma^2)) from
 * Philox4x32-10 keyed (seed, frame) and countered by the global particle id.  With sigma = 0 this
 * is the reference's constant-velocity predict (main.c:875-898) applied to every particle.
 * d_anc (may be NULL) are LOCAL indices into the source arrays (resample gather fused in). */
int slam_motion_sample_dev(slam_engine *e, const float *d_src_x, const float *d_src_y, const float *d_src_th,
                           const int32_t *d_anc, float *d_x, float *d_y, float *d_th, int n, int64_t first_id,
                           const float dp[3], const float sigma[3], uint64_t seed, uint32_t frame);

/* A9 + A7 in one launch: the motion sample above followed by the scan-match score of the new poses
 * (same results as slam_motion_sample_dev then slam_score_poses_dev; source and destination arrays
 * must differ). */
int slam_motion_score_dev(slam_engine *e, int slot, const float *d_src_x, const float *d_src_y,
                          const float *d_src_th, const int32_t *d_anc, float *d_x, float *d_y, float *d_th, int n,
                          int64_t first_id, const float dp[3], const float sigma[3], uint64_t seed, uint32_t frame,
                          float *d_score, int32_t *d_count);

/* A10: per-particle x per-landmark 2x2 EKF correction (FastSLAM 1.0, known correspondences,
 * cartesian sensor-frame observations z = H (m - t), H = [[ct,-st],[st,ct]], noise R = meas_var * I).  H being a rotation
 * and R isotropic, the update is carried out in the world frame: w = t + H^T z, S = P + R, W = P S^-1, mu' = mu + W (w - mu),
 * P' = (I - W) P, log-likelihood term -1/2 (w - mu)^T S^-1 (w - mu) - 1/2 log det S - log 2 pi — the sensor-frame Kalman
 * update in other words, in the fixed operation order of csrc/ekf_math.h == oracle/slam_oracle_pf.c.
 * The observations of the current frame are sensor data like the scan: hand them over once per frame,
 * either as a list from the host (slam_obs_upload_host: landmark ids unique and < nlandmarks, no NaN
 * measurements, nlandmarks <= SLAM_MAX_OBS) or as a table that is already on the device
 * (slam_obs_set_dev).  Inside the engine they are a table indexed by landmark either way.
 * The map is ONE ROW PER PARTICLE: value (plane p, landmark l) of particle i lives at
 * d_map[i * row_stride + p * plane_stride + l], planes mu_x, mu_y, P_xx, P_xy, P_yy, strides in floats
 * (plane_stride >= nlandmarks, row_stride >= 5 * plane_stride; plane_stride a multiple of 32 keeps every
 * row 128-byte aligned and is what the engine's own session uses).  For every observed landmark: read
 * the 5 values of row src(i), update, write them to row i of the output map.  With d_anc != NULL the
 * resample gather is fused in (src(i) = anc[i], a local index; requires d_map_in != d_map_out);
 * landmarks without an observation are copied through when the update is out of place.  P_xx < 0 marks
 * a landmark not seen yet: it is initialised from the observation and contributes no likelihood.
 * The padding columns [nlandmarks, plane_stride) of an output row are unspecified after an out-of-place
 * update (whole 128-landmark batches that fit into the row are processed without predication, padding
 * included, so a plane_stride that is a multiple of 128 is the fastest).
 * loglik[i] (overwritten) is the sum of the observation log-likelihoods in a fixed order: landmark l adds
 * its term to accumulator l mod 128 in order of l (nothing for a landmark without an observation),
 * accumulators j and j+64 are added, and the 64 sums are reduced by a 6-level xor butterfly
 * (t[j] += t[j ^ s], s = 1..32).  The order of the observation list does not matter. */
enum { SLAM_MAX_OBS = 8192 };
int slam_obs_upload_host(slam_engine *e, const int32_t *landmark_id, const float *zx, const float *zy, int nobs,
                         int nlandmarks);
/* Observations already on the device, as the table the engine works on: entry l of the two arrays is the
 * observation of landmark l, NaN in d_zx_by_landmark[l] = landmark l was not observed this frame.  Nothing is
 * copied: the arrays must stay valid until the EKF call that uses them has run.  Their contents may be rewritten
 * between launches without another slam_obs_set_dev: every launch reads them afresh (stream-ordered). */
int slam_obs_set_dev(slam_engine *e, const float *d_zx_by_landmark, const float *d_zy_by_landmark, int nlandmarks);
/* d_loglik may be NULL: the log-likelihoods always stay inside the engine as well, where
 * slam_logweight_ekf_dev picks them up. */
int slam_ekf_update_dev(slam_engine *e, const float *d_map_in, float *d_map_out, int64_t row_stride, int plane_stride,
                        int nlandmarks, const float *d_x, const float *d_y, const float *d_th, const int32_t *d_anc,
                        int n, float meas_var, float *d_loglik);

/* The out-of-place update has two kernels that give the SAME bits: one wavefront per particle (neighbouring particles
 * share their ancestor's row through L2), and one wavefront per 2 or 4 neighbouring particles (the shared row stays in
 * registers).  form = -1 (the initial state): the engine chooses — grouped whenever the update gathers through resample
 * indices, 4 particles per wavefront when its last resample stage reported few distinct ancestors, else 2; 0 forces the
 * first kernel, 1 / 2 the second with 4 / 2 particles per wavefront (tests, measurements; the environment variable
 * SLAM_EKF_GROUP overrides). */
int slam_ekf_form_set(slam_engine *e, int form);
/* out-of-place EKF launches of this engine so far: counts[0] one wavefront per particle, counts[1] the grouped kernel
 * (by itself or inside the fused front launch below) */
int slam_ekf_form_counts(slam_engine *e, int64_t counts[2]);
/* The FRONT of a frame of a single-GPU slam_pf session on rows — motion sample + scan-match score (FastMatch's inner loop,
 * main.c:459-518, for every particle) and the out-of-place landmark update — goes out as ONE launch whose scoring and
 * updating workgroups are dealt out interleaved: the scorer's gathers (texture addresser, L2) run in the shadow of the
 * update's row stores (HBM).  Same bits as the two launches.  on = 1 (the initial state) / 0: two launches (stage timers,
 * measurements; the environment variable SLAM_FRAME_FUSION overrides).  Paged sessions, gated sessions on rows, sharded sessions
 * on rows, short rows and small populations always take the two launches (a sharded session on the split layout fuses the
 * score with the update of the groups whose ancestors are local), and so does every frame while SLAM_PROF_SCORE is being timed
 * (slam_profile_enable: a fused launch is bracketed as SLAM_PROF_EKF).  slam_frame_fusion_count: fused launches of this engine so far. */
int slam_frame_fusion_set(slam_engine *e, int on);
int slam_frame_fusion_count(slam_engine *e, int64_t *launches);
/* Which instantiation the LAST fused front launch of this engine was (tests pin the kernel at the shapes its numbers are
 * quoted on and say which one they pinned): info[0] = particles per updating wavefront (2 or 4), info[1] = lanes per pose of
 * its scoring workgroups (4 below 131 072 particles, else 1); both 0 before the first fused launch. */
int slam_frame_front_last(slam_engine *e, int32_t info[2]);
/* The in-place update (d_map_in == d_map_out: frames that keep their population, slam_resample_gate_set) also has two
 * kernels with the SAME bits: whole rows in batches of 128 landmarks, and the observed landmarks only, from a list
 * the engine compacts out of the observation table once per table (nlandmarks <= 65536).  form = -1 (initial): the
 * second whenever the last list built for this nlandmarks held at most nlandmarks / 4 observations; 0 / 1 force the
 * first / second (tests, measurements; environment variable SLAM_EKF_INPLACE overrides).
 * counts[0] / counts[1]: in-place launches so far of the first / second. */
int slam_ekf_inplace_form_set(slam_engine *e, int form);
int slam_ekf_inplace_form_counts(slam_engine *e, int64_t counts[2]);

/* A11: logw[i] = loglik[i] - score[i] * score_gain  (either input may be NULL = 0) and
 * *d_max = max_i logw[i] (float, device).  Then, with the GLOBAL maximum m (after an all-reduce MAX
 * over shards): wq[i] = (uint64) (exp_det(logw[i] - m) * 2^32)  and  *d_sum = sum(wq)  (exact
 * integer, hence independent of summation order and sharding). */
int slam_logweight_dev(slam_engine *e, const float *d_score, const float *d_loglik, float score_gain, int n,
                       float *d_logw, float *d_max);
int slam_quantise_weights_dev(slam_engine *e, const float *d_logw, const float *d_max, int n, uint64_t *d_wq,
                              uint64_t *d_sum);
/* slam_logweight_dev with loglik = the log-likelihood of the last slam_ekf_update_dev(…, n, …) call on this
 * engine, taken from wherever that call left it (see there). */
int slam_logweight_ekf_dev(slam_engine *e, const float *d_score, float score_gain, int n, float *d_logw,
                           float *d_max);

/* Fused form of the two stages above + the prefix sum below, for the frame loop: weights are quantised and
 * scanned in one pass and never stored; the engine keeps the scan for slam_offspring_from_scan_dev.
 *   d_max  NULL on a single GPU = use the maxima the preceding slam_logweight_dev(…, n, …, d_max = NULL or
 *          not) left inside the engine; on several GPUs pass the all-reduced maximum.
 *   d_sum  shard total written to the device for the all-gather between GPUs; may be NULL on one GPU.
 * slam_logweight_dev accepts d_max == NULL (the global maximum is then not materialised). */
int slam_quantise_scan_dev(slam_engine *e, const float *d_logw, const float *d_max, int n, uint64_t *d_sum);
/* = slam_offspring_offsets_dev on the scan kept by slam_quantise_scan_dev; d_total may be NULL on a
 * single GPU (the grand total is then the shard total, derived inside the kernel). */
int slam_offspring_from_scan_dev(slam_engine *e, int n, const uint64_t *d_base, const uint64_t *d_total,
                                 uint64_t seed, uint32_t frame, int64_t n_total, int32_t *d_first);
/* Several GPUs: base offset and grand total derived inside the kernel from the all-gathered shard totals
 * (d_shard_totals[world], what slam_quantise_scan_dev wrote to d_sum on every rank). */
int slam_offspring_from_scan_sharded_dev(slam_engine *e, int n, const uint64_t *d_shard_totals, int rank, int world,
                                         uint64_t seed, uint32_t frame, int64_t n_total, int32_t *d_first);

/* Resample gate (ESS-gated resampling).  ess_frac in (0, 1): a frame resamples only when the effective sample size
 * of its weights is below ess_frac * N; anything else (the initial state): every frame resamples.  The ESS is taken
 * on 16-bit weights v = wq >> 16, ESS = (sum v)^2 / sum v^2, and compared in exact integer arithmetic
 * ((sum v)^2 * 65536 < round(ess_frac * 65536) * N * sum v^2), so the verdict is the same for any sharding.
 * With a gate set, on this engine:
 *  - slam_quantise_scan_dev also keeps carry[i] = logw[i] - max inside the engine and, when d_sum != NULL, writes
 *    THREE values there: shard total, sum v, sum v^2 (what the ranks all-gather);
 *  - slam_ancestors_from_scan_dev and slam_offspring_from_scan_sharded_dev (whose d_shard_totals then holds such
 *    triples, [world][3]) apply the gate on the device: a frame that keeps its population gets ancestor[j] = j
 *    (first[i] = global index of i), so everything downstream — fused gathers, the exchange plan — works unchanged
 *    and nothing moves;
 *  - slam_logweight_dev / slam_logweight_ekf_dev add the carried weight of the previous frame when that frame did
 *    not resample (a device-side flag; the first frame after slam_resample_gate_set carries nothing);
 *  - slam_resample_happened_host tells the host whether the last gated resample stage did resample (it waits for a
 *    flag in mapped memory — no copy, no stream synchronisation; without a gate the answer is always 1), so that it
 *    can run the next EKF in place (map_in == map_out, no gather) instead of out of place. */
int slam_resample_gate_set(slam_engine *e, float ess_frac);
int slam_resample_happened_host(slam_engine *e, int *resampled);

/* A12: systematic resampling on the exact integer CDF.
 *  step 1 (per shard): d_cdf[i] = inclusive prefix sum of wq within the shard.
 *  step 2 (per shard): d_first[i] = number of comb teeth below the start of particle i's CDF
 *          interval = index of the first output slot it fills; needs the shard's base offset
 *          (sum of wq of all earlier shards; NULL = 0) and the grand total, both read from DEVICE
 *          memory so that the frame loop never synchronises with the host; the comb offset
 *          u = slam_comb_offset(seed, frame, total) is computed inside the kernel.
 *  step 3 (per output slot j of any shard): ancestor[j] = last global i with first[i] <= j, by
 *          binary search in the concatenated `first` array of all shards. */
int slam_prefix_sum_dev(slam_engine *e, const uint64_t *d_wq, int n, uint64_t *d_cdf);
int slam_offspring_offsets_dev(slam_engine *e, const uint64_t *d_cdf, int n, const uint64_t *d_base,
                               const uint64_t *d_total, uint64_t seed, uint32_t frame, int64_t n_total,
                               int32_t *d_first);
int slam_ancestors_dev(slam_engine *e, const int32_t *d_first_all, int64_t n_total, int64_t slot0, int nslots,
                       int32_t *d_anc);
/* Single GPU, frame-loop form: the ancestor of every one of the n slots straight from the scan left by
 * slam_quantise_scan_dev(n) — the same result as slam_offspring_from_scan_dev followed by
 * slam_ancestors_dev, in one launch and without the intermediate `first` array. */
int slam_ancestors_from_scan_dev(slam_engine *e, int n, uint64_t seed, uint32_t frame, int32_t *d_anc);
/* The comb offset of a frame: uniform integer in [0,total) from Philox4x32-10 keyed by seed,
 * counter (0,0,frame,1).  Pure host function (every rank computes the same value). */
uint64_t slam_comb_offset(uint64_t seed, uint32_t frame, uint64_t total);

/* Several GPUs (one engine per rank; n_total = world * n_local particles; world <= 16).  After the `first`
 * arrays of all ranks were all-gathered into d_first_all:
 *  - slam_ancestors_sharded_dev: the gather index of this rank's n_local slots — a local particle index when
 *    the ancestor is local, else n_local + its row in the staging tail behind the local particles.  A remote
 *    ancestor is received ONCE, however many of this rank's slots descend from it: the tail holds, in rank
 *    order and inside a rank in particle order, the distinct remote ancestors of this rank's slots.  Also writes
 *    the exchange plan of the frame to d_plan (SLAM_PLAN_WORDS(world) int32 words, device memory):
 *      [0] bit 0: some rank exchanges something this frame; bit 1: some rank's staging area (see
 *          slam_exchange_set_capacity) might not hold what it would receive — both the same on every rank,
 *      [1 .. world] send_cnt[q]: rows this rank sends to rank q,  [1+world .. 2*world] recv_cnt[q],
 *      [1+2*world .. 3*world] internal to the pack step.
 *    Needs nothing from the host; the host reads the plan once for the all-to-all's split sizes.
 *    d_pose_idx (optional, n_local words): for every slot the index of its ancestor's pose in an all-gather of
 *    the ranks' pose blocks [x[n_local] | y[n_local] | theta[n_local]] — with it the next frame's
 *    slam_motion_score_dev can run on all-gathered poses (d_src_x = gathered, d_src_y = gathered + n_local,
 *    d_src_th = gathered + 2 n_local, d_anc = d_pose_idx) before the exchange of the map rows has happened.
 *  - slam_migrate_pack_dev: one launch packs, for every destination q, its send_cnt[q] rows as records of
 *    3 + 5*nlandmarks floats (x, y, theta, then the five map planes of nlandmarks values each) — the layout of
 *    one all-to-all send buffer.  `plan` is the host copy of d_plan; uses state left by the
 *    slam_ancestors_sharded_dev call of the same frame on this engine.
 *  - slam_migrate_unpack_dev: the received records (recv_cnt[q] from rank q, same layout) into the staging
 *    tail of the pose arrays (rows x|y|theta, leading dimension pose_ld = particle capacity, which the map
 *    must have as rows too) and of the map. */
#define SLAM_PLAN_WORDS(world) (1 + 3 * (world))
int slam_ancestors_sharded_dev(slam_engine *e, const int32_t *d_first_all, int64_t n_total, int n_local, int rank,
                               int world, int32_t *d_src, int32_t *d_plan, int32_t *d_pose_idx);
/* The plan of the last slam_ancestors_sharded_dev call on this engine, on the host: the plan kernel also writes it to
 * pinned host memory mapped into the device and releases an arrival flag; this call waits for that flag (no
 * device-to-host copy, no stream synchronisation) and copies the SLAM_PLAN_WORDS(world) words out. */
int slam_exchange_plan_host(slam_engine *e, int world, int32_t *plan);
/* Rows of staging space behind the local particles (the same on every rank; <= 0 = unlimited, the initial state).
 * With it set, every later plan carries bit 1 of plan[0] when the slots of ANY rank whose ancestor lives on another
 * rank outnumber it — an upper bound of the rows that rank receives, derived from the all-gathered offsets alone, so
 * that all ranks reach the same verdict and can refuse the frame together instead of one rank failing alone. */
int slam_exchange_set_capacity(slam_engine *e, int recv_capacity);
int slam_migrate_pack_dev(slam_engine *e, int n_local, int rank, int world, const int32_t *plan, const float *d_pose,
                          int64_t pose_ld, const float *d_map, int64_t row_stride, int plane_stride, int nlandmarks,
                          float *d_out);
int slam_migrate_unpack_dev(slam_engine *e, const float *d_in, int world, const int32_t *recv_cnt, int n_local,
                            float *d_pose, int64_t pose_ld, float *d_map, int64_t row_stride, int plane_stride,
                            int nlandmarks);

/* Index (lowest on ties) and value of the largest element: the heaviest particle. */
int slam_argmax_dev(slam_engine *e, const float *d_values, int n, int32_t *d_index, float *d_value);

/* Plain gather of particle attributes through an index (used when the gather is not fused into the
 * next stage, and to pack rows for migration between GPUs). */
int slam_gather_f32_dev(slam_engine *e, const float *d_src, const int32_t *d_idx, int n, float *d_dst);
/* map rows: out row i = in row d_idx[i] (strides as in slam_ekf_update_dev) */
int slam_gather_map_dev(slam_engine *e, const float *d_map_in, float *d_map_out, int64_t in_row_stride,
                        int64_t out_row_stride, int in_plane_stride, int out_plane_stride, int nlandmarks,
                        const int32_t *d_idx, int n);

/* ------------------------------------------------------------------ particle-filter session
 * Convenience object for hosts that do not manage device memory themselves (a plain C program): it owns
 * the particle arrays, the landmark maps and the resample indices on the device and runs one whole frame
 * per call by chaining the stage entry points above on the engine's stream — motion+score, EKF (when the
 * filter has landmarks and `use_observations` is set), weights, integer-CDF resample; the resample gather
 * is fused into the next frame.  slam_pf_create: one GPU; slam_pf_create_sharded (below): one session per GPU,
 * every exchange step between the GPUs issued by the engine itself.  The current scan (slam_scan_*), grid (slam_grid_*) and observation list (slam_obs_*) of the
 * engine are the frame's inputs. */
typedef struct slam_pf slam_pf;
typedef struct {
    int32_t n_particles;
    int32_t n_landmarks;   /* 0 = localisation only (weights from the scan-match score alone) */
    float sigma[3];        /* motion noise (x, y, theta) per frame */
    float meas_var;        /* landmark observation variance */
    float score_gain;      /* logw = loglik - score_gain * score */
    uint64_t seed;
    float resample_ess_frac;   /* in (0, 1): resample only when ESS < frac * N (slam_resample_gate_set); 0 = every frame */
    int32_t map_layout;        /* slam_map_layout: how the landmark maps are kept (0 = SLAM_MAP_AUTO) */
} slam_pf_config;
/* Landmark maps: ONE ROW PER PARTICLE (a resampling frame rewrites every row in full: the fastest form when a frame
 * observes most landmarks) or COPY-ON-WRITE PAGES behind a page table per particle (a resampling frame copies table
 * entries and rewrites only the pages that hold an observed landmark: the fastest form when a frame observes few of
 * many).  Both give the same bits.  SLAM_MAP_AUTO lets the session choose and change its mind while it runs. */
typedef enum { SLAM_MAP_AUTO = 0, SLAM_MAP_ROWS = 1, SLAM_MAP_PAGES = 2, SLAM_MAP_SPLIT = 3, SLAM_MAP_SPLIT_PAGES = 4 } slam_map_layout;
/* SLAM_MAP_SPLIT_PAGES (round 4): the split layout with the MEANS on copy-on-write pages of 32 landmarks x 2 planes
 * (256 bytes) and the covariances per class as in SLAM_MAP_SPLIT — what SLAM_MAP_AUTO moves a session to when its
 * frames observe few of many landmarks: a resampling frame copies table entries, rewrites only the mean pages that hold an
 * observed landmark and updates each class's covariances once.  The same bits. */
/* SLAM_MAP_SPLIT (round 4): MEANS per particle, COVARIANCES per covariance class.  The landmark update is carried out in the
 * world frame (see slam_ekf_update_dev): the posterior covariance of a landmark depends on its prior covariance, on meas_var
 * and on whether the frame observes it — never on the particle's pose or on the measurement.  Particles whose covariances are
 * equal therefore stay equal for ever (the offspring of an ancestor; a population whose maps were initialised alike), and
 * they share ONE row of covariance planes, updated once per frame, while every particle keeps its own two planes of means:
 * a resampling frame that observes every landmark reads 8 and writes 8 bytes per (particle, landmark) instead of 20 and 20.
 * Classes are found when maps come in (slam_pf_set_map_*: neighbouring particles with bit-identical covariance planes share a
 * class) and die with their last particle.  The same bits as rows and pages, on one GPU and sharded (classes are local to a
 * rank; a migrating particle becomes a class of its own where it arrives).  slam_pf_device_view gives map = NULL while a
 * session is split, like pages. */
/* SLAM_MAP_AUTO keeps a session on the SPLIT layout instead of rows (single-GPU or sharded, resampling every frame or
 * ESS-gated), and otherwise works as described here, "rows" meaning that dense layout and "pages" SLAM_MAP_SPLIT_PAGES.
 * SLAM_MAP_AUTO starts on rows and watches how many landmarks the frames observe (a count the update kernels leave in
 * mapped host memory: every frame at the start and while they speak against the current layout, every 8th frame otherwise;
 * read without waiting, except in a session's first four frames, which wait for the count of the frame before so that the
 * layout is settled by then): three counts in a row of at most
 * two sevenths of the landmarks move the maps to pages, three in a row of more than three eighths move them back (measured: split pages win
 * below 0.28-0.33 of the landmarks observed, rows and split maps above).  A move costs one
 * pass over the maps and, while it runs, half as much memory again as the steady state (one row buffer beside the pool);
 * when that is not to be had the session stays where it is.  Sessions with at most 32 landmarks stay on rows. */

int slam_pf_create(slam_engine *e, const slam_pf_config *cfg, slam_pf **out);

/* ---- several GPUs: communicators + the sharded session.
 * The reference has no multi-device code; what it does have is ONE C host program that creates
 * its accelerator handle once in main and threads it through (Submodule_2/Hadrware_acclereated.cpp:842-845, 284).
 * The sharded filter keeps that shape: a plain C host creates one engine + one communicator per GPU and then
 * drives every rank with the same slam_pf_* calls as a single GPU; the engine issues every exchange step itself
 * (RCCL over xGMI), on its own streams, with no host framework between the launches.
 *   slam_comm_unique_id      rank 0 makes the rendezvous token (ncclGetUniqueId) and hands it to the other ranks by
 *                            any means (shared memory of the host process, a file, a socket, torch.distributed ...)
 *   slam_comm_create_rccl    collective: every rank calls it with the same token (ncclCommInitRank).  One rank per
 *                            GPU; ranks may be processes or threads of one process.  world <= 16.
 *   slam_local_group_* / slam_comm_create_local
 *                            in-process transport: all ranks are threads of ONE process and exchange through
 *                            device-to-device copies and a host rendezvous.  Same results, no overlap; for hosts that
 *                            do not want RCCL and for rehearsing many ranks on few GPUs (RCCL refuses two ranks on
 *                            one device).  Every rank's calls must come from its own host thread (they block).
 * A communicator belongs to one engine and one session; destroy the session first, then the communicator. */
typedef struct slam_comm slam_comm;
typedef struct slam_local_group slam_local_group;
enum { SLAM_COMM_ID_BYTES = 128 };
int slam_comm_unique_id(uint8_t id[SLAM_COMM_ID_BYTES]);
int slam_comm_create_rccl(slam_engine *e, int rank, int world, const uint8_t id[SLAM_COMM_ID_BYTES], slam_comm **out);
int slam_local_group_create(int world, slam_local_group **out);
int slam_local_group_destroy(slam_local_group *g);
int slam_comm_create_local(slam_engine *e, slam_local_group *g, int rank, slam_comm **out);
/* Give up: after an error of its own a rank aborts its communicator so that the other ranks do not wait for it for
 * ever (ncclCommAbort / the in-process group is marked broken); every later call on it — and, once they notice, on the
 * peers' communicators — returns SLAM_ERR_COMM.  slam_pf_step does this itself when a frame of a sharded session fails
 * on this rank alone.  Host-side waits of a sharded session poll for such failures (ncclCommGetAsyncError) and give
 * up after SLAM_COMM_TIMEOUT_S seconds (environment, default 120). */
int slam_comm_abort(slam_comm *c);
int slam_comm_rank(const slam_comm *c);
int slam_comm_world(const slam_comm *c);
int slam_comm_destroy(slam_comm *c);

/* The sharded session: cfg->n_particles is THIS rank's share (the same on every rank); the population is
 * world x n_particles, particle ids are global (rank * n_particles + local index), and every result is
 * bit-identical to the same population on one GPU.  recv_capacity = rows of staging space for map rows / poses
 * that arrive from other ranks (<= 0: n_particles, which can never overflow; smaller values save memory and make
 * slam_pf_step fail with SLAM_ERR_CAPACITY — on every rank alike — in a frame whose exchange might not fit).
 * Per frame the engine issues, on its own stream and in program order with the kernels around them (no second stream,
 * no event hand-overs): all-reduce MAX of the weight normaliser, all-gather of the shard totals, all-gather of the
 * offspring offsets ("surviving indices"), an all-gather of the new poses (12 B per particle, so that the next frame's
 * motion + score launch need not wait for the exchange), and ONE grouped send/recv carrying the map rows of ancestors
 * that live on another rank (each row once per destination rank).  Every slam_pf_* call on a sharded session is
 * collective. */
int slam_pf_create_sharded(slam_engine *e, const slam_pf_config *cfg, slam_comm *comm, int recv_capacity,
                           slam_pf **out);
int slam_pf_destroy(slam_pf *pf);
/* all particles at `pose`; maps (if any) marked "not seen yet" */
int slam_pf_reset(slam_pf *pf, const float pose[3]);
int slam_pf_set_poses_host(slam_pf *pf, const float *x, const float *y, const float *theta);
int slam_pf_set_map_host(slam_pf *pf, const float *rows /* [n_particles][5][n_landmarks] */);
/* the same from device memory: rows [n_particles][5 planes][plane_stride] floats, row_stride floats apart; asynchronous */
int slam_pf_set_map_dev(slam_pf *pf, const float *d_rows, int64_t row_stride, int plane_stride);
/* Paged maps (SLAM_MAP_PAGES): every particle's landmarks sit behind a page table — pages of 32 landmarks (640 bytes),
 * shared between the offspring of an ancestor until one of them is written: a resample copies 4 bytes per 32 landmarks,
 * and a frame's update rewrites only the pages that hold an observed landmark.  For frames that observe few of many
 * landmarks (a row per particle rewrites every row on every resampling frame); results are bit-identical to a
 * row-per-particle session.  While a session is on pages it has no rows to look at: slam_pf_device_view gives map = NULL,
 * maps go in and out through slam_pf_set_map_* / slam_pf_get_map_host / slam_pf_get_map_rows_host.  Sharded sessions work
 * the same way (a migrating particle travels with all its pages, as a row does).
 * slam_pf_paged_set(e, 1): sessions created on this engine from now on with map_layout = SLAM_MAP_AUTO are kept on pages
 * (as if created with SLAM_MAP_PAGES); 0 restores AUTO's own choice.  slam_pf_is_paged: the layout right now. */
int slam_pf_paged_set(slam_engine *e, int on);
int slam_pf_is_paged(const slam_pf *pf);
/* the layout right now: SLAM_MAP_ROWS, SLAM_MAP_PAGES, SLAM_MAP_SPLIT or SLAM_MAP_SPLIT_PAGES */
int slam_pf_layout(const slam_pf *pf);
/* one frame against grid `slot`; asynchronous */
int slam_pf_step(slam_pf *pf, int slot, const float dp[3], int use_observations);
/* heaviest particle of the last frame (lowest index on ties): its pose, log-weight and index; synchronises.
 * Sharded: the heaviest of the whole population (the same answer on every rank), `index` is its global id. */
int slam_pf_best(slam_pf *pf, float pose[3], float *logw, int32_t *index);
/* Posterior mean of the current (resampled, hence equally weighted) population: x and y are averaged, the heading is
 * averaged on the circle around `ref_theta` (theta = ref + atan2(sum sin(theta_i - ref), sum cos(theta_i - ref)), so a
 * population straddling +-pi does not average to nonsense and theta stays unwrapped — the reference never normalises
 * angles; pass the predicted heading).  The sums are exact fixed-point integer sums made on the device (x, y
 * in 2^-32, sin / cos in 2^-30 units), so the result is bit-identical for any workgroup count and any number of GPUs;
 * one launch, the four sums land in mapped host memory (no copy of the population, no stream synchronisation).
 * Sharded: collective, the same answer on every rank. */
int slam_pf_mean(slam_pf *pf, float ref_theta, float pose[3]);
/* rows this rank received in the exchange of the last completed frame (0 on a single GPU) */
int slam_pf_rows_received(const slam_pf *pf);
/* With cfg.resample_ess_frac in (0, 1): how many of the frames the host has looked at so far did resample (the
 * verdict of a frame is read at the start of the next one).  Without a gate: 0 (every frame resamples). */
int64_t slam_pf_frames_resampled(const slam_pf *pf);
/* how often a SLAM_MAP_AUTO session has moved its maps between rows and pages so far */
int64_t slam_pf_layout_changes(const slam_pf *pf);
/* The session's CURRENT device buffers, for hosts that fill or inspect the population on the device instead of
 * through the *_host copies (a 52 GB map does not want to pass through host memory): pose = [x | y | theta] of
 * n_particles floats each; map = one row per particle, row_stride floats apart, five planes of plane_stride floats
 * (layout of slam_ekf_update_dev); anc = the pending resample gather (slot i descends from particle anc[i] of
 * these buffers; NULL when none is pending: right after create / reset / set_*).  The pointers move with every
 * slam_pf_step; synchronise (slam_engine_sync) before touching the memory from another stream. */
typedef struct {
    float *pose, *map;
    float *map_spare;                 /* the other map buffer: free between frames (the next EKF writes it) */
    const int32_t *anc;
    int64_t row_stride;
    int32_t plane_stride, map_rows;   /* map_rows = n_particles + staging rows (sharded) */
    /* what the last slam_pf_step left behind, indexed like `pose` (i.e. BEFORE the pending gather): the scan-match
     * score, the log-weight and — when that frame used observations, else NULL — the EKF log-likelihood of every
     * particle; NULL before the first frame */
    const float *score, *logw, *loglik;
    const int32_t *count;             /* in-bounds beams behind `score` (main.c:512-518 counts them as bestHits_size) */
} slam_pf_view;
int slam_pf_device_view(slam_pf *pf, slam_pf_view *out);
/* current particles with the pending resample gather applied; synchronises */
int slam_pf_get_poses_host(slam_pf *pf, float *x, float *y, float *theta);
int slam_pf_get_map_host(slam_pf *pf, float *rows /* [n_particles][5][n_landmarks] */);
/* The maps of SOME particles (a host usually wants the heaviest particle's map, not a million of them):
 * rows[k] = the landmarks of current particle particle[k] (0 <= particle[k] < n_particles, repeats allowed), pending
 * gather applied, whatever the layout; synchronises.  Sharded: collective like slam_pf_get_map_host (the exchange of
 * the last frame is completed first), indices are local. */
int slam_pf_get_map_rows_host(slam_pf *pf, const int32_t *particle, int count, float *rows /* [count][5][n_landmarks] */);

/* Inspection of a session that is on pages right now (tests, debugging; SLAM_ERR_NOT_READY when it is on rows).  Device
 * pointers into the session's own state, valid until the next slam_pf_* call that changes the layout:
 *   pool[npages][5][page_landmarks]  the pages;  table[table_rows][pages_per_particle]: page of (particle, block) for the
 *   CURRENT particles before the pending gather (rows n_particles .. table_rows - 1: staging rows of a sharded session);
 *   stamp[npages]: pages named by the current tables carry stamp_now;  freelist[]: entries [state[1], state[0]) are pages
 *   nobody names, entries [state[3], state[1]) were handed out by the last update;  state = {free, used, renewed, base}.
 * SLAM_MAP_SPLIT_PAGES: planes = 2 (a page holds mu_x | mu_y of its 32 landmarks; the covariances are the class's:
 * slam_pf_split_device_view), and the pool is two buffers: page p starts at float offset
 * p * planes * page_landmarks + (p >= half_pages ? gap_floats : 0) from `pool`.  Joint pages: planes = 5, gap_floats = 0. */
typedef struct {
    const float *pool;
    const int32_t *table, *freelist, *state;
    const uint32_t *stamp;
    uint32_t stamp_now;
    int32_t page_landmarks, pages_per_particle, table_rows;
    int64_t npages;
    int32_t planes, reserved;
    int64_t half_pages, gap_floats;
} slam_pf_paged_view;
int slam_pf_paged_device_view(slam_pf *pf, slam_pf_paged_view *out);

/* Inspection of a session that is on the split layout right now (tests, debugging; SLAM_ERR_NOT_READY otherwise).  Device
 * pointers, valid until the next slam_pf_* call: mean[rows][2][plane_stride] and cls[rows] of the CURRENT particles before the
 * pending gather; cov[..][3][plane_stride]: row c = the covariance planes of class c; live[0 .. *live_count): the classes in use
 * as of the last landmark update (a superset of those the current particles name).  SLAM_MAP_SPLIT_PAGES: mean == NULL (the
 * means are on pages: slam_pf_paged_device_view), the rest as described. */
typedef struct {
    const float *mean, *cov;
    const int32_t *cls, *live, *live_count;
    int32_t plane_stride, rows;
} slam_pf_split_view;
int slam_pf_split_device_view(slam_pf *pf, slam_pf_split_view *out);

/* ------------------------------------------------------------------ mapper: the reference's frame loop in one call
 * (SURVEY.md §8f rows N1 + N2).  One slam_mapper_next_frame = one iteration of the reference's loop
 * (main.c:859-969): scan clean-up, world transform, local-map crop, both rasters, both EDTs on key frames,
 * constant-velocity guess, FastMatch + FastMatch2, key-frame test and map append — with the scan, the map,
 * the local map and the grids resident on the device.  Only what the reference computes with libm (cos/sin
 * of beam angles, pose and lattice headings) and the arg-min / key-frame decisions run on the host, so the
 * pose sequence and the final map are bit-identical to the reference program's.  Uses grid slots 0 and 1 of
 * the engine.  `ranges` = one raw scan frame of `nbeams` floats (what main.c:22-30 parses from the CSV). */
typedef struct slam_mapper slam_mapper;
/* The run-time parameters the reference keeps as locals of main() and as literals: one plain struct,
 * slam_mapper_params_default fills in the reference's values. */
typedef struct {
    float fast_res[3];          /* fastResolution  {0.05, 0.05, 0.008727}    main.c:832: lattice step of FastMatch (x = y, theta) */
    float fast_res2[3];         /* fastResolution2 {0.025, 0.025, 0.004363}  main.c:833: ... of FastMatch2 */
    float border;               /* borderSize 1          main.c:834: margin of the local-map crop around the scan, metres */
    float pixel, pixel2;        /* pixelSize 0.2, pixelSize2 0.1   main.c:835-836: coarse / fine grid resolution, metres */
    float key_dt, key_dr;       /* miniUpdateDT 0.3 m, miniUpdateDR 0.0872665 rad   main.c:838-839: key-frame test, per axis */
    float range_min;            /* lidar.range_min 0.023 main.c:50 */
    float usable_range;         /* readAScan(24)         main.c:846, :863 (the reference compares with an int) */
    float edt_cap;              /* 10 cells              main.c:224 (<= 32) */
    float new_point_threshold;  /* 1.5 cells             main.c:943: a beam farther than this from the map is a new map point */
} slam_mapper_params;
void slam_mapper_params_default(slam_mapper_params *p);
/* slam_mapper_create = slam_mapper_create_ex with the defaults (params == NULL means the same) */
int slam_mapper_create(slam_engine *e, int nbeams, float angle_min, float angle_inc, slam_mapper **out);
int slam_mapper_create_ex(slam_engine *e, int nbeams, float angle_min, float angle_inc, const slam_mapper_params *params,
                          slam_mapper **out);
int slam_mapper_destroy(slam_mapper *m);
int slam_mapper_first_frame(slam_mapper *m, const float *ranges);                    /* main.c:844-858 */
int slam_mapper_next_frame(slam_mapper *m, const float *ranges, float pose_out[3]);  /* main.c:859-969 */
/* map points (main.c:982-985 writes them as "%f,%f" lines); x/y may be NULL to query the size only */
int slam_mapper_get_map_host(slam_mapper *m, float *x, float *y, int32_t capacity, int32_t *n);

#ifdef __cplusplus
}
#endif
#endif /* SLAM_HIP_H */
