/*
 * slam_oracle_pf — CPU SPECIFICATION of the particle-filter stages.
 *
 * TEST INFRASTRUCTURE (same rules as slam_oracle.h).
 *
 * PARITY UNPINNED: the reference (circuitpotato/Hardware-Acceleration-of-LIDAR-SLAM) contains no
 * particles, no landmarks, no EKF, no weights and no resampling, so there are
 * no reference outputs to pin these functions to.  They are this build's own specification,
 * checked by analytic known-answer tests (tests/test_oracle_pf.py); the HIP kernels must match them
 * bit for bit.  The only reference anchor is the degenerate case: with zero noise the motion
 * sample is the constant-velocity predict of Subsystem_1/main.c:875-898.
 *
 * Bit-exactness across CPU and GPU is obtained by using only IEEE-754 binary32 +,-,*,/,sqrt
 * (correctly rounded on both), integer arithmetic, and the three "det_" functions below whose
 * evaluation order is part of the specification (no libm, no FMA contraction).
 */
#ifndef SLAM_ORACLE_PF_H
#define SLAM_ORACLE_PF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- specified elementary functions (DESIGN.md "deterministic math") */
void orc_det_sincosf(float a, float *s, float *c);
float orc_det_expf(float x);   /* x <= 0 expected; returns 0 below -80 */
float orc_det_logf(float x);   /* x > 0, normal */
void orc_det_sincosf_array(const float *a, int n, float *s, float *c);
void orc_det_expf_array(const float *x, int n, float *y);
void orc_det_logf_array(const float *x, int n, float *y);

/* ---- counter-based RNG: Philox4x32-10 (Salmon et al., SC'11), published constants */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* ---- scan-match score with the specified device trig (particle mode of row A7) */
struct orc_grid_meta_s;
void orc_score_poses_det(const void *grid_meta, const float *edt, const float *bx, const float *by, int nbeams,
                         const float *x, const float *y, const float *theta, int nposes, float *score,
                         int32_t *count);

/* ---- A9 motion sample */
void orc_motion_sample(const float *src_x, const float *src_y, const float *src_th, const int32_t *anc, float *x,
                       float *y, float *th, int n, int64_t first_id, const float dp[3], const float sigma[3],
                       uint64_t seed, uint32_t frame);

/* ---- A10 per-particle x per-landmark 2x2 EKF.  The map is one row per particle: value (plane p, landmark l) of
 * particle i at map[i * row_stride + p * plane_stride + l], planes mu_x mu_y P_xx P_xy P_yy.
 * Log-likelihood summation order (part of the specification): the landmarks are walked by index; landmark l
 * adds its term to accumulator l mod 128 (+0.0f when it has no observation, on a first sighting, and for the
 * slots that pad the count to a multiple of 128); accumulators j and j+64 are added; the 64 sums are reduced
 * by a 6-level xor butterfly t[j] = t[j] + t[j ^ s], s = 1, 2, 4, 8, 16, 32.  The order of the observation
 * list does not matter (landmark ids in it are unique). */
enum { ORC_EKF_LANES = 128 };
void orc_ekf_update(const float *map_in, float *map_out, int64_t row_stride, int plane_stride, int nlandmarks,
                    const float *x, const float *y, const float *th, const int32_t *anc, int n,
                    const int32_t *obs_id, const float *obs_zx, const float *obs_zy, int nobs, float meas_var,
                    float *loglik);

/* ---- A11 weights */
void orc_logweight(const float *score, const float *loglik, float score_gain, int n, float *logw, float *max_out);
void orc_quantise_weights(const float *logw, float max, int n, uint64_t *wq, uint64_t *sum);

/* ---- A11/A12 resample gate (ESS-gated resampling).  The effective sample size is taken on 16-bit weights
 * v_i = wq_i >> 16 (so that every sum fits 64 bits for up to 2^31 particles):  S = sum v_i,  Q = sum v_i^2,
 * ESS = S^2 / Q.  The population is resampled in a frame iff  ESS < frac * N  with frac = frac_q16 / 65536, decided in
 * exact integer arithmetic:  S^2 * 65536 < frac_q16 * N * Q  (128-bit products).  Both sums are plain integer sums,
 * hence independent of summation order and of how the particles are sharded.
 * A frame that does NOT resample keeps every particle in its slot (ancestor = itself) and carries its weight into
 * the next frame: carry_i = logw_i - max (one float subtraction), and that frame's log-weight is
 * carry_i + (loglik_i - gain*score_i) (one more float addition). */
void orc_ess_terms(const uint64_t *wq, int n, uint64_t *s16, uint64_t *q16);
int orc_ess_resample(uint64_t s16, uint64_t q16, int64_t n_total, uint32_t frac_q16);
void orc_logweight_carry(const float *score, const float *loglik, float score_gain, const float *carry, int n,
                         float *logw, float *max_out);
void orc_weight_carry(const float *logw, float max, int n, float *carry);

/* ---- A12 systematic resample on the integer CDF */
void orc_prefix_sum(const uint64_t *wq, int n, uint64_t *cdf);
uint64_t orc_comb_offset(uint64_t seed, uint32_t frame, uint64_t total);
void orc_offspring_offsets(const uint64_t *cdf, int n, uint64_t base, uint64_t total, uint64_t comb_u,
                           int64_t n_total, int32_t *first);
void orc_ancestors(const int32_t *first_all, int64_t n_total, int64_t slot0, int nslots, int32_t *anc);

/* test helper for the scorer's rounding identity (see slam_oracle_pf.c) */
uint64_t orc_round_trick_mismatches(uint32_t lo, uint32_t hi, uint32_t step);

#ifdef __cplusplus
}
#endif
#endif
