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

// What the update of a landmark that HAS been seen before gives (o0..o4 = mu_x, mu_y, P_xx, P_xy, P_yy; ll = its
// log-likelihood term) and what a first sighting gives (f0, f1: the observed point in the world frame; P = q I, no term).
// The caller selects: prior P_xx < 0 -> first sighting; no observation -> the prior values, no term.
template <class T> struct EkfResult {
    T o0, o1, o2, o3, o4, ll, f0, f1;
};

// what a first sighting puts into the map: the observed point in the world frame
template <class T> __device__ __forceinline__ void ekf_first_sighting(T zx, T zy, T s, T c, T px, T py, T& f0, T& f1)
{
    f0 = px + (c * zx + s * zy);
    f1 = py + (c * zy - s * zx);
}

// prior (mx, my, pxx, pxy, pyy), observation (zx, zy) in the sensor frame, pose (px, py, heading sine s / cosine c), q = R.
// WITH_FIRST = false leaves f0 / f1 unset: the row kernels work them out (ekf_first_sighting) only for a batch of landmarks
// that holds a first sighting at all, which in a running filter is rare.
template <class T, bool WITH_FIRST = true>
__device__ __forceinline__ EkfResult<T> ekf_update_one(T mx, T my, T pxx, T pxy, T pyy, T zx, T zy, T s, T c, T px, T py, T q)
{
    EkfResult<T> r;
    const T dx = mx - px, dy = my - py;
    const T vx = zx - (c * dx - s * dy);
    const T vy = zy - (s * dx + c * dy);
    const T a00 = c * pxx - s * pxy, a01 = c * pxy - s * pyy;
    const T a10 = s * pxx + c * pxy, a11 = s * pxy + c * pyy;
    const T s00 = (a00 * c - a01 * s) + q;
    const T s01 = a00 * s + a01 * c;
    const T s11 = (a10 * s + a11 * c) + q;
    const T det = s00 * s11 - s01 * s01;
    const T idet = ekf_rcp(det);
    const T i00 = s11 * idet, i01 = -s01 * idet, i11 = s00 * idet;
    const T k00 = a00 * i00 + a10 * i01, k01 = a00 * i01 + a10 * i11;
    const T k10 = a01 * i00 + a11 * i01, k11 = a01 * i01 + a11 * i11;
    r.o0 = mx + (k00 * vx + k01 * vy);
    r.o1 = my + (k10 * vx + k11 * vy);
    r.o2 = pxx - (k00 * a00 + k01 * a10);
    r.o3 = pxy - (k00 * a01 + k01 * a11);
    r.o4 = pyy - (k10 * a01 + k11 * a11);
    const T maha = vx * (i00 * vx + i01 * vy) + vy * (i01 * vx + i11 * vy);
    r.ll = ((ekf_splat<T>(0.0f) - ekf_splat<T>(0.5f) * maha) - ekf_splat<T>(0.5f) * ekf_log(det)) - ekf_splat<T>(1.8378770664f);
    if (WITH_FIRST) ekf_first_sighting<T>(zx, zy, s, c, px, py, r.f0, r.f1);
    return r;
}

}  // namespace slam
