// det_math.h — device-side "specified" elementary functions and Philox4x32-10 for gfx950.
//
// The particle-filter stages have no reference implementation, so
// their numerics are specified by this build (DESIGN.md "deterministic math").  To make every
// stage bit-reproducible between the GPU and the CPU checker, and independent of how particles
// are sharded over GPUs, only correctly-rounded IEEE binary32 operations are used: each multiply
// and add below is a separate rounding (the translation units are compiled with
// -ffp-contract=off), in exactly the order written.  Polynomial coefficients are the classic
// single-precision Cephes minimax sets.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slam {

// round half away from zero, as C roundf (the reference's cell selection, main.c:483,501)
__device__ __forceinline__ float round_half_away(float v)
{
    const float t = truncf(v);
    const float d = fabsf(v - t);              // exact
    return d >= 0.5f ? t + copysignf(1.0f, v) : t;
}

__device__ __forceinline__ void det_sincosf(float a, float& s, float& c)
{
    const float kf = round_half_away(a * 0.636619772f);
    const int k = (int)kf;
    float r = a - kf * 1.5703125f;
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
    const int quad = k & 3;
    const float s0 = (quad & 1) ? cr : sr;
    const float c0 = (quad & 1) ? sr : cr;
    s = (quad & 2) ? -s0 : s0;
    c = ((quad + 1) & 2) ? -c0 : c0;
}

__device__ __forceinline__ float det_expf(float x)
{
    if (!(x > -80.0f)) return 0.0f;
    if (x > 0.0f) x = 0.0f;
    const float kf = round_half_away(x * 1.44269504f);
    const int k = (int)kf;
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
    return p * __uint_as_float((uint32_t)(k + 127) << 23);
}

__device__ __forceinline__ float det_logf(float x)
{
    if (!(x >= 1.17549435e-38f)) x = 1.17549435e-38f;
    const uint32_t u = __float_as_uint(x);
    int e = (int)(u >> 23) - 126;
    const float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
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

struct u32x4 { uint32_t v[4]; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1)
{
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0;
        const uint32_t n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u32x4{ { c0, c1, c2, c3 } };
}

// Motion sample of one particle (row A9; specification: oracle/slam_oracle_pf.c orc_motion_sample).
struct MotionParams {
    float dp[3];
    float sigma[3];
    uint32_t key0, key1, frame;
    uint64_t first_id;
};

inline MotionParams make_motion_params(int64_t first_id, const float dp[3], const float sigma[3], uint64_t seed,
                                       uint32_t frame)
{
    MotionParams mp;
    for (int k = 0; k < 3; ++k) {
        mp.dp[k] = dp[k];
        mp.sigma[k] = sigma[k];
    }
    mp.key0 = (uint32_t)seed;
    mp.key1 = (uint32_t)(seed >> 32);
    mp.frame = frame;
    mp.first_id = (uint64_t)first_id;
    return mp;
}

__device__ __forceinline__ void motion_sample_one(const MotionParams& mp, uint64_t local_index, float sx, float sy,
                                                  float sth, float& x, float& y, float& th)
{
    const uint64_t gid = mp.first_id + local_index;
    const u32x4 r = philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), mp.frame, 0u /* motion stream */, mp.key0,
                                  mp.key1);
    const float u1 = (float)((r.v[0] >> 8) + 1u) * 5.9604644775390625e-8f;
    const float u2 = (float)(r.v[1] >> 8) * 5.9604644775390625e-8f;
    const float u3 = (float)((r.v[2] >> 8) + 1u) * 5.9604644775390625e-8f;
    const float u4 = (float)(r.v[3] >> 8) * 5.9604644775390625e-8f;
    float s1, c1, s2, c2;
    const float rad1 = sqrtf(-2.0f * det_logf(u1));   // sqrtf: correctly rounded (NOT __fsqrt_rn)
    det_sincosf(6.2831853072f * u2, s1, c1);
    const float rad2 = sqrtf(-2.0f * det_logf(u3));
    det_sincosf(6.2831853072f * u4, s2, c2);
    const float z0 = rad1 * c1, z1 = rad1 * s1, z2 = rad2 * c2;
    x = (sx + mp.dp[0]) + mp.sigma[0] * z0;
    y = (sy + mp.dp[1]) + mp.sigma[1] * z1;
    th = (sth + mp.dp[2]) + mp.sigma[2] * z2;
}

}  // namespace slam
