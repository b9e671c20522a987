// ekf_math.h — the arithmetic of ONE landmark update (no counterpart in the reference, specification:
// oracle/slam_oracle_pf.c orc_ekf_update), written once for every kernel that applies it: the row walk, the grouped row
// walk, the compact observation list (pf_kernels.hip) and the paged update (paged_kernels.hip).  T = float (one landmark per
// lane) or v2f (two landmarks per lane on packed arithmetic: v_pk_mul_f32 / v_pk_add_f32 are IEEE per component, so both
// give the same bits).  Every multiply and add is rounded separately, in this order (-ffp-contract=off).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "det_math.h"

namespace slam {

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f bc2(float a) { return (v2f){a, a}; }

// det_logf on two values: the integer steps per component, the polynomial packed (same operation order)
__device__ __forceinline__ v2f det_logf2(v2f x)
{
    float m_[2], ef_[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float xv = x[t];
        if (!(xv >= 1.17549435e-38f)) xv = 1.17549435e-38f;
        const uint32_t u = __float_as_uint(xv);
        int e = (int)(u >> 23) - 126;
        const float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
        const bool lo = m < 0.70710678f;
        e = lo ? e - 1 : e;
        m_[t] = lo ? m + m : m;
        ef_[t] = (float)e;
    }
    const v2f f = (v2f){m_[0], m_[1]} - bc2(1.0f), ef = (v2f){ef_[0], ef_[1]};
    const v2f z = f * f;
    v2f y = bc2(7.0376836292e-2f) * f;
    y = y + bc2(-1.1514610310e-1f); y = y * f;
    y = y + bc2(1.1676998740e-1f);  y = y * f;
    y = y + bc2(-1.2420140846e-1f); y = y * f;
    y = y + bc2(1.4249322787e-1f);  y = y * f;
    y = y + bc2(-1.6668057665e-1f); y = y * f;
    y = y + bc2(2.0000714765e-1f);  y = y * f;
    y = y + bc2(-2.4999993993e-1f); y = y * f;
    y = y + bc2(3.3333331174e-1f);  y = y * f;
    y = y * z;
    y = y + ef * bc2(-2.12194440e-4f);
    y = y - bc2(0.5f) * z;
    v2f r = f + y;
    r = r + ef * bc2(0.693359375f);
    return r;
}

template <class T> __device__ __forceinline__ T ekf_splat(float a);
template <> __device__ __forceinline__ float ekf_splat<float>(float a) { return a; }
template <> __device__ __forceinline__ v2f ekf_splat<v2f>(float a) { return bc2(a); }
// 1 / d, correctly rounded (the specification says 1.0f / det: IEEE division, what the CPU computes).  hipcc's expansion of an
// IEEE float division is v_div_scale x2, v_rcp, a Newton step, two quotient corrections with v_div_fmas, v_div_fixup — eleven
// instructions per value, a fifth of the landmark update's arithmetic after the logarithm.  When the numerator is 1 and the
// denominator's exponent lies in [-60, 60] neither scale nor fix-up does anything, and what remains is the reciprocal
// estimate and six fused multiply-adds — the same operations on the same values, hence the same bits
// (slam_selftest_reciprocal walks every float of that range on the device and compares with the compiler's division:
// tests/test_gpu_pf.py).  Outside the range (wave-uniform test) the division itself.
__device__ __forceinline__ bool ekf_rcp_in_range(float d)
{
    return ((__float_as_uint(d) >> 23) & 0xffu) - 67u <= 120u;   // 2^-60 <= |d| < 2^61, finite
}
__device__ __forceinline__ float ekf_rcp_core(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float t = __builtin_fmaf(-d, r, 1.0f);
    float q = __builtin_fmaf(t, r, r);
    t = __builtin_fmaf(-d, q, 1.0f);
    return __builtin_fmaf(t, r, q);
}
// (SLAM_EKF_PLAIN_DIV: measurement builds with the division itself everywhere)
__device__ __forceinline__ float ekf_rcp(float d)
{
#ifndef SLAM_EKF_PLAIN_DIV
    if (__ballot(!ekf_rcp_in_range(d)) == 0) return ekf_rcp_core(d);
#endif
    return 1.0f / d;
}
__device__ __forceinline__ v2f ekf_rcp(v2f d)
{
#ifndef SLAM_EKF_PLAIN_DIV
    if (__ballot(!(ekf_rcp_in_range(d[0]) && ekf_rcp_in_range(d[1]))) == 0) {
        v2f r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        const v2f one = {1.0f, 1.0f}, nd = -d;
        const v2f e = __builtin_elementwise_fma(nd, r, one);
        r = __builtin_elementwise_fma(e, r, r);
        v2f t = __builtin_elementwise_fma(nd, r, one);
        v2f q = __builtin_elementwise_fma(t, r, r);
        t = __builtin_elementwise_fma(nd, q, one);
        return __builtin_elementwise_fma(t, r, q);
    }
#endif
    return (v2f){1.0f / d[0], 1.0f / d[1]};
}
__device__ __forceinline__ float ekf_log(float d) { return det_logf(d); }
__device__ __forceinline__ v2f ekf_log(v2f d) { return det_logf2(d); }

// ---- The update of one landmark, in the WORLD frame (specification: oracle/slam_oracle_pf.c orc_ekf_update).
// The observation model is z = H (m - t) with H a rotation and R = q I, so
//     S_sensor = H P H^T + q I = H (P + q I) H^T,   K = P H^T S_sensor^-1 = P (P + q I)^-1 H^T = W H^T,
//     K nu = W (H^T nu) = W (w - mu)   with w = t + H^T z the observed point in the world frame,
//     P' = (I - K H) P = (I - W) P,    nu^T S_sensor^-1 nu = (w - mu)^T (P + q I)^-1 (w - mu),   det S_sensor = det (P + q I):
// everything that involves the covariance — S^-1, the gain W, P', the determinant's logarithm — depends on the prior P and
// on q ALONE, not on the particle's pose and not on the measurement.  The offspring of one ancestor share its P, so a
// wavefront that updates several of them (the grouped kernels) works that part out ONCE per landmark; per particle there
// remain the observed point w, the innovation d = w - mu, mu' = mu + W d and the Mahalanobis term.  Round 3 replaced the
// sensor-frame formulation (the same algebra with H carried through every product: ~160 vector instructions per landmark
// pair and particle) by this one: ~75 shared + ~31 per particle.

// the part that depends on the prior covariance and q only
template <class T> struct EkfShared {
    T i00, i01, i11;        // (P + q I)^-1
    T w00, w01, w10, w11;   // gain in the world frame, W = P (P + q I)^-1
    T o2, o3, o4;           // posterior covariance (I - W) P: P_xx, P_xy, P_yy
    T hl;                   // 0.5 * log det (P + q I)
};

// ... in two steps.  The two expensive values — the reciprocal of det (P + q I) and half its logarithm — and then the rest.
// The split layout (split_kernels.hip) keeps the two values beside every covariance class's planes: they are worked out
// when the class's covariances are, once, and the particles' update starts from them (ekf_shared_from).
template <class T> __device__ __forceinline__ void ekf_det_terms(T pxx, T pxy, T pyy, T q, T& idet, T& hl)
{
    const T a = pxx + q, c = pyy + q;
    const T det = a * c - pxy * pxy;
    idet = ekf_rcp(det);
    hl = ekf_splat<T>(0.5f) * ekf_log(det);
}

// WITH_POSTERIOR = false: o2 .. o4 are not wanted (the particles' update of the split layout)
template <class T, bool WITH_POSTERIOR = true>
__device__ __forceinline__ EkfShared<T> ekf_shared_from(T pxx, T pxy, T pyy, T q, T idet, T hl)
{
    EkfShared<T> h;
    const T a = pxx + q, c = pyy + q;
    h.i00 = c * idet;
    h.i01 = -pxy * idet;
    h.i11 = a * idet;
    h.w00 = pxx * h.i00 + pxy * h.i01;
    h.w01 = pxx * h.i01 + pxy * h.i11;
    h.w10 = pxy * h.i00 + pyy * h.i01;
    h.w11 = pxy * h.i01 + pyy * h.i11;
    if constexpr (WITH_POSTERIOR) {
        h.o2 = pxx - (h.w00 * pxx + h.w01 * pxy);
        h.o3 = pxy - (h.w00 * pxy + h.w01 * pyy);
        h.o4 = pyy - (h.w10 * pxy + h.w11 * pyy);
    }
    h.hl = hl;
    return h;
}

template <class T> __device__ __forceinline__ EkfShared<T> ekf_shared(T pxx, T pxy, T pyy, T q)
{
    T idet, hl;
    ekf_det_terms<T>(pxx, pxy, pyy, q, idet, hl);
    return ekf_shared_from<T>(pxx, pxy, pyy, q, idet, hl);
}

// the part that depends on the particle: its pose (px, py, heading sine s / cosine c), the prior mean and the measurement
template <class T> struct EkfParticle {
    T o0, o1;   // posterior mean
    T ll;       // log-likelihood term
    T wx, wy;   // the observed point in the world frame (what a first sighting stores)
};

template <class T>
__device__ __forceinline__ EkfParticle<T> ekf_particle(const EkfShared<T>& h, T mx, T my, T zx, T zy, T s, T c, T px, T py)
{
    EkfParticle<T> r;
    r.wx = px + (c * zx + s * zy);
    r.wy = py + (c * zy - s * zx);
    const T dx = r.wx - mx, dy = r.wy - my;
    r.o0 = mx + (h.w00 * dx + h.w01 * dy);
    r.o1 = my + (h.w10 * dx + h.w11 * dy);
    const T maha = dx * (h.i00 * dx + h.i01 * dy) + dy * (h.i01 * dx + h.i11 * dy);
    r.ll = ((ekf_splat<T>(0.0f) - ekf_splat<T>(0.5f) * maha) - h.hl) - ekf_splat<T>(1.8378770664f);
    return r;
}

// Both parts for one landmark and one particle.  What the update of a landmark that HAS been seen before gives (o0..o4 =
// mu_x, mu_y, P_xx, P_xy, P_yy; ll = its log-likelihood term) and what a first sighting gives (f0, f1: the observed point
// in the world frame; P = q I, no term).  The caller selects: prior P_xx < 0 -> first sighting; no observation -> the prior
// values, no term.
template <class T> struct EkfResult {
    T o0, o1, o2, o3, o4, ll, f0, f1;
};

// what a first sighting puts into the map: the observed point in the world frame
template <class T> __device__ __forceinline__ void ekf_first_sighting(T zx, T zy, T s, T c, T px, T py, T& f0, T& f1)
{
    f0 = px + (c * zx + s * zy);
    f1 = py + (c * zy - s * zx);
}

// prior (mx, my, pxx, pxy, pyy), observation (zx, zy) in the sensor frame, pose (px, py, heading sine s / cosine c), q = R
template <class T, bool WITH_FIRST = true>
__device__ __forceinline__ EkfResult<T> ekf_update_one(T mx, T my, T pxx, T pxy, T pyy, T zx, T zy, T s, T c, T px, T py, T q)
{
    const EkfShared<T> h = ekf_shared<T>(pxx, pxy, pyy, q);
    const EkfParticle<T> u = ekf_particle<T>(h, mx, my, zx, zy, s, c, px, py);
    EkfResult<T> r;
    r.o0 = u.o0;
    r.o1 = u.o1;
    r.o2 = h.o2;
    r.o3 = h.o3;
    r.o4 = h.o4;
    r.ll = u.ll;
    r.f0 = u.wx;   // the same expression as ekf_first_sighting
    r.f1 = u.wy;
    return r;
}

}  // namespace slam
