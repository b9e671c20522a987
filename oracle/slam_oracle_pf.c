/*
 * slam_oracle_pf.c — CPU specification of the particle-filter stages (rows A9-A12).
 * TEST INFRASTRUCTURE; PARITY UNPINNED (no reference counterpart) — see slam_oracle_pf.h.
 *
 * Only the zero-noise motion step has a reference anchor: Subsystem_1/main.c:875-898
 * (pose_guess = pose + (pose - previous_pose)).
 */
#include "slam_oracle_pf.h"
#include "slam_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ deterministic math
 * Polynomials are the classic single-precision Cephes minimax sets; what is SPECIFIED here is
 * the exact sequence of binary32 operations (each multiply and add rounded on its own). */

static inline float f_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t bits_from_f(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

void orc_det_sincosf(float a, float *s, float *c)
{
    const int k = (int)roundf(a * 0.636619772f);   /* nearest multiple of pi/2 */
    const float kf = (float)k;
    float r = a - kf * 1.5703125f;                 /* three-part Cody-Waite reduction */
    r = r - kf * 4.837512969970703125e-4f;
    r = r - kf * 7.549789948768648e-8f;
    const float r2 = r * r;
    float p = -1.9515295891e-4f * r2;
    p = p + 8.3321608736e-3f;
    p = p * r2;
    p = p + -1.6666654611e-1f;
    p = p * r2;
    p = p * r;
    const float sr = p + r;
    float q = 2.443315711809948e-5f * r2;
    q = q + -1.388731625493765e-3f;
    q = q * r2;
    q = q + 4.166664568298827e-2f;
    q = q * r2;
    q = q * r2;
    q = q - 0.5f * r2;
    const float cr = q + 1.0f;
    switch (k & 3) {
    case 0: *s = sr; *c = cr; break;
    case 1: *s = cr; *c = -sr; break;
    case 2: *s = -sr; *c = -cr; break;
    default: *s = -cr; *c = sr; break;
    }
}

float orc_det_expf(float x)
{
    if (!(x > -80.0f)) return 0.0f;   /* also catches NaN */
    if (x > 0.0f) x = 0.0f;
    const int k = (int)roundf(x * 1.44269504f);
    const float kf = (float)k;
    float r = x - kf * 0.693359375f;
    r = r - kf * -2.12194440e-4f;
    const float r2 = r * r;
    float p = 1.9875691500e-4f * r;
    p = p + 1.3981999507e-3f;
    p = p * r;
    p = p + 8.3334519073e-3f;
    p = p * r;
    p = p + 4.1665795894e-2f;
    p = p * r;
    p = p + 1.6666665459e-1f;
    p = p * r;
    p = p + 5.0000001201e-1f;
    p = p * r2;
    p = p + r;
    p = p + 1.0f;
    return p * f_from_bits((uint32_t)(k + 127) << 23);
}

float orc_det_logf(float x)
{
    if (!(x >= 1.17549435e-38f)) x = 1.17549435e-38f;
    const uint32_t u = bits_from_f(x);
    int e = (int)(u >> 23) - 126;
    const float m = f_from_bits((u & 0x007fffffu) | 0x3f000000u);   /* [0.5, 1) */
    float f;
    if (m < 0.70710678f) {
        e = e - 1;
        f = (m + m) - 1.0f;
    } else {
        f = m - 1.0f;
    }
    const float z = f * f;
    float y = 7.0376836292e-2f * f;
    y = y + -1.1514610310e-1f;
    y = y * f;
    y = y + 1.1676998740e-1f;
    y = y * f;
    y = y + -1.2420140846e-1f;
    y = y * f;
    y = y + 1.4249322787e-1f;
    y = y * f;
    y = y + -1.6668057665e-1f;
    y = y * f;
    y = y + 2.0000714765e-1f;
    y = y * f;
    y = y + -2.4999993993e-1f;
    y = y * f;
    y = y + 3.3333331174e-1f;
    y = y * f;
    y = y * z;
    const float ef = (float)e;
    y = y + ef * -2.12194440e-4f;
    y = y - 0.5f * z;
    float r = f + y;
    r = r + ef * 0.693359375f;
    return r;
}

void orc_det_sincosf_array(const float *a, int n, float *s, float *c)
{
    for (int i = 0; i < n; ++i) orc_det_sincosf(a[i], &s[i], &c[i]);
}
void orc_det_expf_array(const float *x, int n, float *y)
{
    for (int i = 0; i < n; ++i) y[i] = orc_det_expf(x[i]);
}
void orc_det_logf_array(const float *x, int n, float *y)
{
    for (int i = 0; i < n; ++i) y[i] = orc_det_logf(x[i]);
}

/* ------------------------------------------------------------------ Philox4x32-10 */

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ------------------------------------------------------------------ A7, particle mode */

void orc_score_poses_det(const void *grid_meta, const float *edt, const float *bx, const float *by, int nbeams,
                         const float *x, const float *y, const float *theta, int nposes, float *score,
                         int32_t *count)
{
    const orc_grid_meta *g = (const orc_grid_meta *)grid_meta;
    for (int p = 0; p < nposes; ++p) {
        float s, c;
        int n;
        orc_det_sincosf(theta[p], &s, &c);
        score[p] = orc_score_pose(g, edt, bx, by, nbeams, x[p], y[p], c, s, NULL, &n);
        count[p] = n;
    }
}

/* ------------------------------------------------------------------ A9 */

enum { STREAM_MOTION = 0, STREAM_RESAMPLE = 1 };

void orc_motion_sample(const float *src_x, const float *src_y, const float *src_th, const int32_t *anc, float *x,
                       float *y, float *th, int n, int64_t first_id, const float dp[3], const float sigma[3],
                       uint64_t seed, uint32_t frame)
{
    const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    for (int i = 0; i < n; ++i) {
        const uint64_t gid = (uint64_t)(first_id + i);
        const uint32_t ctr[4] = { (uint32_t)gid, (uint32_t)(gid >> 32), frame, STREAM_MOTION };
        uint32_t r[4];
        orc_philox4x32_10(ctr, key, r);
        /* Box-Muller on 24-bit uniforms: u in (0,1] for the radius, [0,1) for the angle */
        const float u1 = (float)((r[0] >> 8) + 1u) * 5.9604644775390625e-8f;
        const float u2 = (float)(r[1] >> 8) * 5.9604644775390625e-8f;
        const float u3 = (float)((r[2] >> 8) + 1u) * 5.9604644775390625e-8f;
        const float u4 = (float)(r[3] >> 8) * 5.9604644775390625e-8f;
        float s1, c1, s2, c2;
        const float rad1 = sqrtf(-2.0f * orc_det_logf(u1));
        orc_det_sincosf(6.2831853072f * u2, &s1, &c1);
        const float rad2 = sqrtf(-2.0f * orc_det_logf(u3));
        orc_det_sincosf(6.2831853072f * u4, &s2, &c2);
        (void)s2;
        const float z0 = rad1 * c1, z1 = rad1 * s1, z2 = rad2 * c2;
        const int j = anc ? anc[i] : i;
        const float nx = (src_x[j] + dp[0]) + sigma[0] * z0;
        const float ny = (src_y[j] + dp[1]) + sigma[1] * z1;
        const float nt = (src_th[j] + dp[2]) + sigma[2] * z2;
        x[i] = nx;
        y[i] = ny;
        th[i] = nt;
    }
}

/* ------------------------------------------------------------------ A10 */

void orc_ekf_update(const float *map_in, float *map_out, int64_t row_stride, int plane_stride, int nlandmarks,
                    const float *x, const float *y, const float *th, const int32_t *anc, int n,
                    const int32_t *obs_id, const float *obs_zx, const float *obs_zy, int nobs, float meas_var,
                    float *loglik)
{
    const float q = meas_var;
    /* observation of landmark l, if any (ids are unique) */
    int *obs_of = (int *)malloc(sizeof(int) * (size_t)(nlandmarks > 0 ? nlandmarks : 1));
    for (int l = 0; l < nlandmarks; ++l) obs_of[l] = -1;
    for (int k = 0; k < nobs; ++k) obs_of[obs_id[k]] = k;
    /* landmark slots are padded to whole groups of 128; a padded or unobserved slot contributes +0.0f */
    const int nslots = (nlandmarks + ORC_EKF_LANES - 1) / ORC_EKF_LANES * ORC_EKF_LANES;
    const size_t ps = (size_t)plane_stride;
    for (int i = 0; i < n; ++i) {
        const int src = anc ? anc[i] : i;
        const float *in = map_in + (size_t)src * row_stride;
        float *out = map_out + (size_t)i * row_stride;
        float st, ct;
        orc_det_sincosf(th[i], &st, &ct);
        const float px = x[i], py = y[i];
        float lane[ORC_EKF_LANES];
        for (int j = 0; j < ORC_EKF_LANES; ++j) lane[j] = 0.0f;
        for (int l = 0; l < nslots; ++l) {
            float ll = 0.0f;
            const int k = l < nlandmarks ? obs_of[l] : -1;
            if (l < nlandmarks && k < 0) {
                /* no observation this frame: gathered copy when the update is out of place */
                if (map_in != map_out)
                    for (int p = 0; p < 5; ++p) out[p * ps + l] = in[p * ps + l];
            } else if (k >= 0) {
                const float mx = in[l], my = in[ps + l], pxx = in[2 * ps + l], pxy = in[3 * ps + l], pyy = in[4 * ps + l];
                const float zx = obs_zx[k], zy = obs_zy[k];
                if (pxx < 0.0f) {
                    /* first sighting: place the landmark at the observed point, P = R; no likelihood term */
                    out[l] = px + (ct * zx + st * zy);
                    out[ps + l] = py + (ct * zy - st * zx);
                    out[2 * ps + l] = q;
                    out[3 * ps + l] = 0.0f;
                    out[4 * ps + l] = q;
                } else {
                    /* Observation model z = H (mu - t), H = [[ct,-st],[st,ct]] (inverse of the reference's R^T, main.c:115-116),
                     * R = q I.  H is a rotation, so the update is carried out in the WORLD frame, where everything that
                     * involves the covariance is independent of the pose: S = P + q I, W = P S^-1, P' = (I - W) P,
                     * mu' = mu + W (w - mu) with w = t + H^T z the observed point, nu^T S_sensor^-1 nu = (w-mu)^T S^-1 (w-mu),
                     * det S_sensor = det S (hardware-acceleration-of-lidar-slam_amd/csrc/ekf_math.h: the same operations) */
                    const float wx = px + (ct * zx + st * zy);
                    const float wy = py + (ct * zy - st * zx);
                    const float a = pxx + q, c = pyy + q;
                    const float det = a * c - pxy * pxy;
                    const float idet = 1.0f / det;
                    const float i00 = c * idet, i01 = -pxy * idet, i11 = a * idet;         /* S^-1 */
                    const float w00 = pxx * i00 + pxy * i01, w01 = pxx * i01 + pxy * i11;   /* W = P S^-1 */
                    const float w10 = pxy * i00 + pyy * i01, w11 = pxy * i01 + pyy * i11;
                    const float dx = wx - mx, dy = wy - my;
                    out[l] = mx + (w00 * dx + w01 * dy);
                    out[ps + l] = my + (w10 * dx + w11 * dy);
                    out[2 * ps + l] = pxx - (w00 * pxx + w01 * pxy);                      /* (I - W) P */
                    out[3 * ps + l] = pxy - (w00 * pxy + w01 * pyy);
                    out[4 * ps + l] = pyy - (w10 * pxy + w11 * pyy);
                    const float maha = dx * (i00 * dx + i01 * dy) + dy * (i01 * dx + i11 * dy);
                    const float hl = 0.5f * orc_det_logf(det);
                    ll = ((0.0f - 0.5f * maha) - hl) - 1.8378770664f;
                }
            }
            /* summation order (the specification): landmark l goes to accumulator l mod 128, in order of l ... */
            lane[l % ORC_EKF_LANES] = lane[l % ORC_EKF_LANES] + ll;
        }
        /* ... then accumulators j and j+64 are added, then a 6-level xor butterfly over the 64 sums */
        float t[64];
        for (int j = 0; j < 64; ++j) t[j] = lane[j] + lane[j + 64];
        for (int s = 1; s < 64; s <<= 1) {
            float u[64];
            for (int j = 0; j < 64; ++j) u[j] = t[j] + t[j ^ s];
            for (int j = 0; j < 64; ++j) t[j] = u[j];
        }
        loglik[i] = t[0];
    }
    free(obs_of);
}

/* ------------------------------------------------------------------ A11 */

void orc_logweight(const float *score, const float *loglik, float score_gain, int n, float *logw, float *max_out)
{
    float m = -INFINITY;
    for (int i = 0; i < n; ++i) {
        const float ll = loglik ? loglik[i] : 0.0f;
        const float sc = score ? score[i] * score_gain : 0.0f;
        const float lw = ll - sc;
        logw[i] = lw;
        if (lw > m) m = lw;
    }
    *max_out = m;
}

void orc_quantise_weights(const float *logw, float max, int n, uint64_t *wq, uint64_t *sum)
{
    uint64_t s = 0;
    for (int i = 0; i < n; ++i) {
        const float w = orc_det_expf(logw[i] - max);
        wq[i] = (uint64_t)(w * 4294967296.0f);
        s += wq[i];
    }
    *sum = s;
}

/* ------------------------------------------------------------------ resample gate */

void orc_ess_terms(const uint64_t *wq, int n, uint64_t *s16, uint64_t *q16)
{
    uint64_t s = 0, q = 0;
    for (int i = 0; i < n; ++i) {
        const uint64_t v = wq[i] >> 16;
        s += v;
        q += v * v;
    }
    *s16 = s;
    *q16 = q;
}

int orc_ess_resample(uint64_t s16, uint64_t q16, int64_t n_total, uint32_t frac_q16)
{
    const unsigned __int128 lhs = ((unsigned __int128)s16 * s16) << 16;
    const unsigned __int128 rhs = (unsigned __int128)q16 * ((uint64_t)n_total * (uint64_t)frac_q16);
    return lhs < rhs;
}

void orc_logweight_carry(const float *score, const float *loglik, float score_gain, const float *carry, int n,
                         float *logw, float *max_out)
{
    float m = -INFINITY;
    for (int i = 0; i < n; ++i) {
        const float ll = loglik ? loglik[i] : 0.0f;
        const float sc = score ? score[i] * score_gain : 0.0f;
        float lw = ll - sc;
        if (carry) lw = carry[i] + lw;
        logw[i] = lw;
        if (lw > m) m = lw;
    }
    *max_out = m;
}

void orc_weight_carry(const float *logw, float max, int n, float *carry)
{
    for (int i = 0; i < n; ++i) carry[i] = logw[i] - max;
}

/* ------------------------------------------------------------------ A12 */

void orc_prefix_sum(const uint64_t *wq, int n, uint64_t *cdf)
{
    uint64_t s = 0;
    for (int i = 0; i < n; ++i) {
        s += wq[i];
        cdf[i] = s;
    }
}

uint64_t orc_comb_offset(uint64_t seed, uint32_t frame, uint64_t total)
{
    const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    const uint32_t ctr[4] = { 0, 0, frame, STREAM_RESAMPLE };
    uint32_t r[4];
    orc_philox4x32_10(ctr, key, r);
    const uint64_t r64 = (uint64_t)r[0] | ((uint64_t)r[1] << 32);
    return (uint64_t)(((unsigned __int128)r64 * total) >> 64);   /* uniform in [0, total) */
}

void orc_offspring_offsets(const uint64_t *cdf, int n, uint64_t base, uint64_t total, uint64_t comb_u,
                           int64_t n_total, int32_t *first)
{
    /* comb tooth j sits at j*total + u on an axis where particle i spans [N*C(i-1), N*C(i));
     * first[i] = number of teeth strictly below N*C(i-1) = ceil((N*C(i-1) - u) / total), clamped at 0 */
    for (int i = 0; i < n; ++i) {
        const uint64_t c_excl = base + (i ? cdf[i - 1] : 0);
        const unsigned __int128 X = (unsigned __int128)c_excl * (uint64_t)n_total;
        first[i] = X <= comb_u ? 0 : (int32_t)((X - comb_u - 1) / total + 1);
    }
}

void orc_ancestors(const int32_t *first_all, int64_t n_total, int64_t slot0, int nslots, int32_t *anc)
{
    /* ancestor of slot j = last particle whose first slot is <= j (particles with no offspring share
     * their successor's first slot and are skipped by taking the LAST one) */
    for (int s = 0; s < nslots; ++s) {
        const int64_t j = slot0 + s;
        int64_t lo = 0, hi = n_total;   /* first index with first_all[idx] > j */
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)first_all[mid] <= j) lo = mid + 1; else hi = mid;
        }
        anc[s] = (int32_t)(lo - 1);
    }
}

/* ------------------------------------------------------------------ test helper: the scorer's cell rounding
 * The HIP scorer computes (int)roundf(v) as trunc(v + copysignf(0.5f - 1 ulp, v)).  This walks float bit patterns
 * [lo, hi] (step `step`) and counts where that differs from libm's roundf — it must be 0 over all 2^32 patterns. */
uint64_t orc_round_trick_mismatches(uint32_t lo, uint32_t hi, uint32_t step)
{
    uint64_t bad = 0;
    for (uint64_t u = lo; u <= hi; u += step) {
        const float v = f_from_bits((uint32_t)u);
        if (v != v) continue;   /* NaN: both forms convert to 0 on the device */
        const float a = truncf(v + copysignf(0x1.fffffep-2f, v));
        const float b = roundf(v);
        if (a != b) ++bad;
    }
    return bad;
}
