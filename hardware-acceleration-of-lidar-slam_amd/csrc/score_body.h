// score_body.h — the scan-match score of a slice of poses as a device function (semantics restated from
// Subsystem_1/main.c:381-596, see score_kernels.hip): the body of score_poses_kernel, shared with the fused front kernel of
// a particle-filter frame (pf_kernels.hip: scoring workgroups and landmark-update workgroups in ONE launch).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "det_math.h"
#include "kernels.h"

namespace slam {

constexpr int kScoreBlock = 256;

// LPP = lanes per pose.
//  LPP 1: one lane walks all beams of its pose.  Best when there are enough poses to fill the chip (>= ~128k).
//  LPP 4: a quad of lanes shares a pose, lane q takes beams 4j+q.  The reference's sequential float sum
//         is kept exactly: after each step every lane of the quad adds the four hits in beam order,
//         fetched with DPP quad broadcasts (v_add_f32 ... quad_perm:[k,k,k,k]) — no LDS, no extra
//         rounding.  4x the wavefronts and 4x the gathers in flight: 1.5x faster at 64k poses, where
//         LPP 1 has a single wave per SIMD and is latency-bound; slower beyond ~128k poses (each wave
//         gather then touches 4 beam neighbourhoods instead of 1).
// Both forms run the same software pipeline: DEPTH gathers stay in flight per lane and the hits of step s are summed
// while the loads of the following steps are outstanding (left to itself hipcc waits vmcnt(0) behind every gather:
// the 32-fold unrolled loop this replaces had ONE gather in flight per wavefront).
// What the inner loop costs (rocprofv3 counters at 1M poses x 360 beams on a 2048^2 grid, profiles/): the texture
// addresser (one cache line per cycle: ~60 distinct lines per wave gather for unordered poses) and the vector ALU
// (it was 32 instructions per beam) are both near their limits, HBM is idle.  Hence, per beam:
//  - two beams per step on float2 (v_pk_mul_f32 / v_pk_add_f32: IEEE per component, the scalar bits);
//  - (int)roundf(v) as trunc(v + copysign(0.5 - 1ulp, v)): one bfi, one add and the (truncating) conversion instead of
//    trunc / subtract / compare / select / add / trunc — exact for every float (tests/test_oracle_pf.py walks all 2^32);
//  - the cell offset as one 24-bit multiply-add (v_mad_u32_u24 is full rate, v_mul_lo_u32 is not);
//  - the EDT as a buffer resource: an out-of-bounds beam gets offset 0xffffffff and the hardware's range check
//    returns +0.0f, which leaves the running sum's bits unchanged — no select on the loaded value, no 64-bit address.
// PACKED (round 4): the gathers go to a one-byte-per-cell copy of the grid laid out in 16-column strips (kernels.h:
// ScoreGrid::packed), a byte is turned back into the cell's float through a 256-entry table in LDS (the same bits: the
// capped EDT only holds 0, sqrtf of small integers and the cap, main.c:223-269).  Why: the scorer is bound by the texture
// addresser, which works through the DISTINCT 128-byte lines of a wave gather one per cycle (~59 of them for the 64 lanes of
// a settled population on a row-major float grid, profiles/r02_pmc_score.md: the poses of a wavefront spread over ~30 rows,
// and every row is another line).  In the packed copy a line is a 16 x 8 patch of cells, so the same neighbourhood lies in
// a fraction of the lines.  An out-of-bounds beam still gets offset 0xffffffff: the range check returns byte 0 = the code of
// an occupied cell = +0.0f.  The decode costs one LDS read per beam, issued one beam pair ahead of the sum that consumes it.
constexpr int kQuadDepth = 8;    // gathers in flight per lane, 4 lanes per pose (measured: 43 -> 39 us at 64k poses x 360
                                 // beams, 84 -> 47 us at 16k x 1079)
constexpr int kLaneDepth = 16;   // gathers in flight per lane, 1 lane per pose

template <int CTRL>
__device__ __forceinline__ float quad_bcast(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// MOTION: the pose is not read but produced — pose' = motion_sample(src[anc[i]]) (row A9) — written to
// (px,py,p2) by the pose's first lane and scored in the same launch (saves a launch and a pose round trip).
// `block`: which 256-thread slice of the poses this workgroup takes (blockIdx.x of score_poses_kernel; the fused front kernel
// of a frame, pf_kernels.hip, hands its scoring workgroups their slice); s_pair: nb_pad / 2 float4 of LDS.
template <bool HAS_CS, int LPP, int DEPTH, bool MOTION, bool PACKED = false>
__device__ __forceinline__ void score_poses_body(const ScoreGrid& g, const float* __restrict__ bx, const float* __restrict__ by,
                                                 int nbeams, float* __restrict__ px, float* __restrict__ py,
                                                 float* __restrict__ p2, const float* __restrict__ p3, int nposes,
                                                 float* __restrict__ score, int32_t* __restrict__ count, const MotionIO& mio,
                                                 const MotionParams& mpar, int block, float4* s_pair)
{
    // beams padded with NaN to a whole number of pipeline rounds: a NaN beam is out of bounds and adds +0
    // (at least one round, so that an empty scan still runs the pipeline prologue on NaN beams)
    constexpr int kRound = LPP * DEPTH;
    const int nb_pad = nbeams > 0 ? (nbeams + kRound - 1) / kRound * kRound : kRound;
    {
        // LPP lanes per pose: lane `sub` takes beams sub + LPP*k and handles them two at a time (k = 2m, 2m+1) on
        // float2 arithmetic; the pair is staged side by side — slot sub + LPP*m = (x_k, x_k+1, y_k, y_k+1) — so one
        // 16-byte LDS read (a broadcast: every lane of a wave reads the same slot) delivers both operands packed
        const float nanv = __builtin_nanf("");
        for (int p = threadIdx.x; p < nb_pad / 2; p += kScoreBlock) {
            const int m = p / LPP, su = p - m * LPP;
            const int b0 = su + LPP * (2 * m), b1 = b0 + LPP;
            s_pair[p] = make_float4(b0 < nbeams ? bx[b0] * g.ipix : nanv, b1 < nbeams ? bx[b1] * g.ipix : nanv,
                                    b0 < nbeams ? by[b0] * g.ipix : nanv, b1 < nbeams ? by[b1] * g.ipix : nanv);
        }
    }
    // PACKED: the decode table behind the beams (the launcher adds its 1 KB to the dynamic LDS size)
    float* s_table = reinterpret_cast<float*>(s_pair + nb_pad / 2);
    if constexpr (PACKED) {
        static_assert(kScoreBlock == 256, "one table entry per thread");
        s_table[threadIdx.x] = g.table[threadIdx.x];
    }
    __syncthreads();

    const int t = block * kScoreBlock + threadIdx.x;
    const int pose = t / LPP, sub = t % LPP;
    if (LPP == 1 && pose >= nposes) return;
    const int i = pose < nposes ? pose : nposes - 1;   // LPP > 1: keep whole quads active for the DPP broadcasts

    float ct, st, pose_x, pose_y;
    if constexpr (MOTION) {
        const int j = mio.anc ? mio.anc[i] : i;
        float pose_t;
        motion_sample_one(mpar, (uint64_t)i, mio.sx[j], mio.sy[j], mio.sth[j], pose_x, pose_y, pose_t);
        if (sub == 0 && pose < nposes) {
            px[i] = pose_x;
            py[i] = pose_y;
            p2[i] = pose_t;
        }
        det_sincosf(pose_t, st, ct);
    } else {
        pose_x = px[i];
        pose_y = py[i];
        if (HAS_CS) {
            ct = p2[i];
            st = p3[i];
        } else {
            det_sincosf(p2[i], st, ct);
        }
    }
    const float nst = -st;
    const float off_x = (pose_x - g.min_x) * g.ipix;
    const float off_y = (pose_y - g.min_y) * g.ipix;
    // The reference's test (int)roundf(v) + 1 > 1 && ... + 1 < n is 1 <= c <= n - 2 on the rounded cell c, i.e.
    // (unsigned)(c - 1) < n - 2: one subtract and one unsigned compare per axis, no branches.  The conversion
    // saturates and maps NaN to 0, so far-away and padding (NaN) beams fail the test like they fail the float one.
    const unsigned lim_x = (unsigned)(g.cols > 2 ? g.cols - 2 : 0);
    const unsigned lim_y = (unsigned)(g.rows > 2 ? g.rows - 2 : 0);
    const unsigned ld4 = (unsigned)g.ld * 4u;   // < 2^24 (the engine refuses wider grids), rows < 2^24: 24-bit multiply
    const unsigned strip = (unsigned)g.strip_bytes;   // PACKED: bytes of one 16-column strip (< 2^24)
    const __amdgpu_buffer_rsrc_t edt =
        PACKED ? __builtin_amdgcn_make_buffer_rsrc((void*)g.packed, 0, (int)(((unsigned)g.cols + 15u) / 16u * strip), 0x00020000)
               : __builtin_amdgcn_make_buffer_rsrc((void*)g.edt, 0, (int)((unsigned)g.rows * ld4), 0x00020000);

    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f c2 = {ct, ct}, s2 = {st, st}, ns2 = {nst, nst}, ox2 = {off_x, off_x}, oy2 = {off_y, off_y};
    float total = 0.0f;
    int n_in = 0;
    typedef int v2i __attribute__((ext_vector_type(2)));
    const v2i sign2 = {(int)0x80000000, (int)0x80000000}, half2 = {0x3effffff, 0x3effffff};   // 0.5 - 1 ulp
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    auto beam2 = [&](int m) {   // beams sub + LPP*2m and sub + LPP*(2m+1); PACKED: the two cells' codes (bit patterns)
        const float4 q = s_pair[sub + LPP * m];
        const v2f X = {q.x, q.y}, Y = {q.z, q.w};
        v2f fx = ((X * c2) + (Y * s2)) + ox2;
        v2f fy = ((X * ns2) + (Y * c2)) + oy2;
        // (int)roundf(f), half away from zero (the reference's cell selection, main.c:483, 501), as
        // trunc(f + copysign(0.5 - 1 ulp, f)) on both beams at once: the sum is exact or rounds to the right side of the
        // integer for every float (checked over all 2^32 patterns, tests/test_oracle_pf.py); the conversion below
        // truncates, saturates and maps NaN to 0 like the (int)roundf() of the float-valued form it replaces
        fx = fx + __builtin_bit_cast(v2f, (__builtin_bit_cast(v2i, fx) & sign2) | half2);
        fy = fy + __builtin_bit_cast(v2f, (__builtin_bit_cast(v2i, fy) & sign2) | half2);
        v2f h;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int ix, iy;   // the conversion truncates by itself (__float2int_rz puts a v_trunc_f32 in front of it)
            asm("v_cvt_i32_f32 %0, %1" : "=v"(ix) : "v"(fx[e]));
            asm("v_cvt_i32_f32 %0, %1" : "=v"(iy) : "v"(fy[e]));
            const bool in = (unsigned)(ix - 1) < lim_x && (unsigned)(iy - 1) < lim_y;
            if constexpr (PACKED) {
                const unsigned off = in ? __umul24((unsigned)ix >> 4, strip) + (((unsigned)iy << 4) | ((unsigned)ix & 15u)) : 0xffffffffu;
                h[e] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b8(edt, (int)off, 0, 0));   // out of range: code 0
            } else {
                const unsigned off = in ? __umul24((unsigned)iy, ld4) + ((unsigned)ix << 2) : 0xffffffffu;
                h[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(edt, (int)off, 0, 0));   // out of range: +0.0f
            }
            n_in += in ? 1 : 0;
        }
        return h;
    };
    auto decode = [&](v2f c) {   // PACKED: code -> float through the LDS table
        const v2u k = __builtin_bit_cast(v2u, c);
        return (v2f){s_table[k[0]], s_table[k[1]]};
    };
    auto add_hit = [&](float h) {
        if constexpr (LPP == 4) {   // beams 4j, 4j+1, 4j+2, 4j+3 in order, identically in all four lanes of the quad
            total = total + quad_bcast<0x00>(h);
            total = total + quad_bcast<0x55>(h);
            total = total + quad_bcast<0xAA>(h);
            total = total + quad_bcast<0xFF>(h);
        } else {
            total = total + h;
        }
    };
    constexpr int kPairs = DEPTH / 2;
    v2f hq[kPairs];
#pragma unroll
    for (int m = 0; m < kPairs; ++m) hq[m] = beam2(m);
    if constexpr (PACKED) {
        // three stages: gathers DEPTH beams ahead, the LDS decode one pair ahead, the ordered sum.  (The sum starts by adding
        // the two zeros of `hprev`: +0 + +0 = +0, the bits of the reference's sum.)
        v2f hprev = {0.0f, 0.0f};
        for (int r = 1; r < nb_pad / kRound; ++r) {
#pragma unroll
            for (int m = 0; m < kPairs; ++m) {
                const v2f c = hq[m];
                hq[m] = beam2(r * kPairs + m);
                const v2f h = decode(c);
                add_hit(hprev[0]);
                add_hit(hprev[1]);
                hprev = h;
            }
        }
#pragma unroll
        for (int m = 0; m < kPairs; ++m) {
            const v2f h = decode(hq[m]);
            add_hit(hprev[0]);
            add_hit(hprev[1]);
            hprev = h;
        }
        add_hit(hprev[0]);
        add_hit(hprev[1]);
    } else {
        for (int r = 1; r < nb_pad / kRound; ++r) {
#pragma unroll
            for (int m = 0; m < kPairs; ++m) {
                const v2f h = hq[m];
                hq[m] = beam2(r * kPairs + m);
                add_hit(h[0]);
                add_hit(h[1]);
            }
        }
#pragma unroll
        for (int m = 0; m < kPairs; ++m) {
            add_hit(hq[m][0]);
            add_hit(hq[m][1]);
        }
    }
    if (LPP == 4) {
        n_in += __builtin_amdgcn_mov_dpp(n_in, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
        n_in += __builtin_amdgcn_mov_dpp(n_in, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    }
    if (sub == 0 && pose < nposes) {
        score[pose] = total;
        count[pose] = n_in;
    }
}

// Pose-count threshold below which the 4-lanes-per-pose form wins (measured on MI355X, 360 beams, pipelined kernels of
// round 2: 64k poses 28.5 vs 30.1 us; 128k poses 55.4 vs 53.2 us; 256k poses 109 vs 104 us).
constexpr int kQuadMaxPoses = 131072;
// ... and below which one wavefront per pose wins over the quad form (360 / 1079 beams: 2k poses 6.1 vs 7.9 us / 18.3 vs
// 24.4 us; 4k poses 9.2 vs 8.0 us / 27.8 vs 24.6 us; 8k poses 14.9 vs 8.0 us / 44.1 vs 25.1 us).  The quad form is flat
// up to 8k poses: there its time is the chain of beam steps, not the work.
constexpr int kWaveMaxPoses = 3072;

}  // namespace slam
