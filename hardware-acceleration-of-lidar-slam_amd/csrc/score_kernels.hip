// score_kernels.hip — scan-match score of many poses against the capped EDT.
//
// Semantics restated from Subsystem_1/main.c:381-596 (FastMatch) and SURVEY.md Appendix A.5:
//   per pose (x, y, theta):  off = ((x - min_x) * ipix, (y - min_y) * ipix)
//   per beam b, in order:    q = (bx*ipix, by*ipix)            main.c:417-421
//                            r = (qx*ct + qy*st, qx*(-st) + qy*ct)   main.c:462-463 (R^T convention)
//                            S = (int)roundf(r + off) + 1      main.c:483, :501
//                            if 1 < Sx < nCols and 1 < Sy < nRows:   main.c:512
//                                score += EDT[Sy-1][Sx-1]      main.c:515-516 (sequential float sum)
// The reference evaluates this for the 27 poses of a lattice; a particle filter evaluates the very
// same function for N arbitrary poses, so one kernel serves both.
//
// Mapping to CDNA4: one lane = one pose, so the float sum runs in the reference's beam order and is
// bit-exact (a tree reduction over beams would change low bits and can flip the arg-min).
// The beams are shared by every pose: they are pixel-scaled once per workgroup and staged in LDS;
// inside the loop every lane reads the same LDS address (a broadcast, conflict-free).  The EDT
// (4-16 MiB) is gathered straight from L2/Infinity Cache; neighbouring lanes hold neighbouring poses
// (particles are sorted by ancestor after resampling) so one wave's 64 gathers fall in a few lines.
// Compiled with -ffp-contract=off: every multiply and add below is rounded separately.

#include <stdlib.h>

#include "det_math.h"
#include "free_list_body.h"
#include "kernels.h"
#include "score_body.h"

namespace slam {

namespace {

template <bool HAS_CS, int LPP, int DEPTH, bool MOTION, bool PACKED>
__global__ __launch_bounds__(kScoreBlock) void score_poses_kernel(ScoreGrid g, const float* __restrict__ bx,
                                                                   const float* __restrict__ by, int nbeams,
                                                                   float* __restrict__ px, float* __restrict__ py,
                                                                   float* __restrict__ p2,
                                                                   const float* __restrict__ p3, int nposes,
                                                                   float* __restrict__ score,
                                                                   int32_t* __restrict__ count, MotionIO mio,
                                                                   MotionParams mpar)
{
    extern __shared__ float4 s_pair[];
    if constexpr (MOTION) {   // the workgroups behind the scorer's: a paged session's free list (kernels.h: FreeListRider)
        if (mio.rider.stamp && (int)blockIdx.x >= mio.rider.first_block) {
            free_list_body(mio.rider.stamp, mio.rider.npages, mio.rider.live, mio.rider.freelist, mio.rider.pool_state, mio.rider.h_short,
                           (int)blockIdx.x - mio.rider.first_block, mio.rider.nblocks);
            return;
        }
    }
    score_poses_body<HAS_CS, LPP, DEPTH, MOTION, PACKED>(g, bx, by, nbeams, px, py, p2, p3, nposes, score, count, mio, mpar,
                                                         (int)blockIdx.x, s_pair);
}

// Small batches (< ~8k poses): ONE WAVEFRONT PER POSE.  The 64 lanes gather 64 beams at a time (8 x 64 in flight),
// park the hits — 0 for out-of-bounds beams, which leaves the sum's bits unchanged — in LDS in beam order, and
// lane 0 adds them up in that order.  19 vs 45 us at 1k poses x 1024 beams, 30 vs 45 us at 4k; slower than the
// quad form from ~8k poses on.
template <bool HAS_CS, bool MOTION>
__global__ __launch_bounds__(kScoreBlock) void score_poses_wave_kernel(ScoreGrid g, const float* __restrict__ bx,
                                                                        const float* __restrict__ by, int nbeams,
                                                                        float* __restrict__ px, float* __restrict__ py,
                                                                        float* __restrict__ p2,
                                                                        const float* __restrict__ p3, int nposes,
                                                                        float* __restrict__ score,
                                                                        int32_t* __restrict__ count, MotionIO mio,
                                                                        MotionParams mpar)
{
    extern __shared__ float s_wave_hits[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * (kScoreBlock / 64) + wave;
    if (i >= nposes) return;   // whole waves leave together; no workgroup barrier below
    float* __restrict__ sh = s_wave_hits + (size_t)wave * nbeams;
    float ct, st, pose_x, pose_y;
    if constexpr (MOTION) {
        const int j = mio.anc ? mio.anc[i] : i;
        float pose_t;
        motion_sample_one(mpar, (uint64_t)i, mio.sx[j], mio.sy[j], mio.sth[j], pose_x, pose_y, pose_t);
        if (lane == 0) {
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
    const float lim_x = (float)(g.cols - 1);
    const float lim_y = (float)(g.rows - 1);
    int n_in = 0;
    constexpr int kGroup = 8;
    for (int b0 = 0; b0 < nbeams; b0 += 64 * kGroup) {
        float qx[kGroup], qy[kGroup], h[kGroup];
        bool in[kGroup];
#pragma unroll
        for (int k = 0; k < kGroup; ++k) {
            const int b = b0 + 64 * k + lane;
            const int bb = b < nbeams ? b : 0;
            qx[k] = bx[bb];
            qy[k] = by[bb];
        }
#pragma unroll
        for (int k = 0; k < kGroup; ++k) {
            const int b = b0 + 64 * k + lane;
            const float x = qx[k] * g.ipix, y = qy[k] * g.ipix;
            const float rx = (x * ct) + (y * st);
            const float ry = (x * nst) + (y * ct);
            const float fx = round_half_away(rx + off_x);
            const float fy = round_half_away(ry + off_y);
            in[k] = (b < nbeams) & (fx > 0.0f) & (fy > 0.0f) & (fx < lim_x) & (fy < lim_y);
            h[k] = g.edt[in[k] ? (int)fy * g.ld + (int)fx : 0];
        }
#pragma unroll
        for (int k = 0; k < kGroup; ++k) {
            const int b = b0 + 64 * k + lane;
            if (b < nbeams) sh[b] = in[k] ? h[k] : 0.0f;
            n_in += __popcll(__ballot(in[k]));
        }
    }
    // hand-off inside one wavefront: drain this wave's LDS writes, then lane 0 reads them
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        float total = 0.0f;
#pragma unroll 8
        for (int b = 0; b < nbeams; ++b) total = total + sh[b];
        score[i] = total;
        count[i] = n_in;
    }
}

// One pose, hits written compacted in beam order (FastMatchParameters.bestHits, main.c:515).
// Single wave: each step handles 64 consecutive beams; ballot + lane prefix give the in-order slot.
__global__ __launch_bounds__(64) void pose_hits_kernel(ScoreGrid g, const float* __restrict__ bx,
                                                       const float* __restrict__ by, int nbeams,
                                                       const float* __restrict__ pose_xycs,
                                                       float* __restrict__ hits, int32_t* __restrict__ count)
{
    const float x = pose_xycs[0], y = pose_xycs[1], ct = pose_xycs[2], st = pose_xycs[3];
    const float nst = -st;
    const float off_x = (x - g.min_x) * g.ipix;
    const float off_y = (y - g.min_y) * g.ipix;
    const float lim_x = (float)(g.cols - 1);
    const float lim_y = (float)(g.rows - 1);
    const int lane = threadIdx.x;
    int base = 0;
    for (int b0 = 0; b0 < nbeams; b0 += 64) {
        const int b = b0 + lane;
        bool in = false;
        float h = 0.0f;
        if (b < nbeams) {
            const float qx = bx[b] * g.ipix, qy = by[b] * g.ipix;
            const float rx = (qx * ct) + (qy * st);
            const float ry = (qx * nst) + (qy * ct);
            const float fx = round_half_away(rx + off_x);
            const float fy = round_half_away(ry + off_y);
            in = (fx > 0.0f) & (fy > 0.0f) & (fx < lim_x) & (fy < lim_y);
            if (in) h = g.edt[(int)fy * g.ld + (int)fx];
        }
        const unsigned long long mask = __ballot(in);
        const int rank = __popcll(mask & ((1ull << lane) - 1ull));
        if (in) hits[base + rank] = h;
        base += __popcll(mask);
    }
    if (lane == 0) *count = base;
}

constexpr int kLatticeN = 27;

// One wavefront per lattice candidate.  Beams are taken 64 at a time: every lane classifies one beam,
// ballot + lane prefix give the beam's in-order slot among the in-bounds ones, the hit goes to LDS
// (for the ordered sum) and to this candidate's row of `work` (for the merge below).  The score is
// then accumulated by lane 0 in slot order — the reference's sequential float sum (main.c:516).
__global__ __launch_bounds__(64) void lattice_kernel(ScoreGrid g, const float* __restrict__ bx,
                                                     const float* __restrict__ by, int nbeams_max,
                                                     const int32_t* __restrict__ d_nbeams,
                                                     const float* __restrict__ cand, float* __restrict__ work,
                                                     float* __restrict__ out, const float* __restrict__ prev_out,
                                                     const float* __restrict__ pair_in)
{
    extern __shared__ float s_hit[];
    const int c = blockIdx.x, lane = threadIdx.x;
    // the beam count may live on the device (scan cleaned up there); rows of `work` keep the max stride
    int nbeams = d_nbeams ? *d_nbeams : nbeams_max;
    nbeams = nbeams < nbeams_max ? nbeams : nbeams_max;
    float x = cand[c], y = cand[kLatticeN + c], ct = cand[2 * kLatticeN + c], st = cand[3 * kLatticeN + c];
    if (prev_out) {
        // the second call of a chained pair (kernels.h: launch_lattice_pair): `cand` is the FIRST call's table, this candidate is
        // laid out around that call's winner (main.c:549-563: strict '<' keeps the first of equal scores)
        float best = INFINITY;
        int bk = -1;
        for (int k = 0; k < kLatticeN; ++k)
            if (prev_out[k] < best) {
                best = prev_out[k];
                bk = k;
            }
        if (bk < 0) bk = kLatticeN / 2;   // nothing below +inf: the input pose = the middle candidate (heading 1, x 1, y 1)
        const float wx = cand[bk], wy = cand[kLatticeN + bk], t = pair_in[18];
        const int a = bk / 9, b = c / 9, i = (c / 3) % 3, j = c % 3;
        x = i == 0 ? wx - t : (i == 1 ? wx : wx + t);
        y = j == 0 ? wy - t : (j == 1 ? wy : wy + t);
        ct = pair_in[a * 3 + b];
        st = pair_in[9 + a * 3 + b];
    }
    const float nst = -st;
    const float off_x = (x - g.min_x) * g.ipix;
    const float off_y = (y - g.min_y) * g.ipix;
    const float lim_x = (float)(g.cols - 1);
    const float lim_y = (float)(g.rows - 1);
    float* __restrict__ row = work + (size_t)c * nbeams_max;
    int base = 0;
    constexpr int kGroup = 8;   // 8 x 64 beams: all their gathers are in flight before the first is consumed
    for (int b0 = 0; b0 < nbeams; b0 += 64 * kGroup) {
        float h[kGroup], qx[kGroup], qy[kGroup];
        bool in[kGroup];
        // phase 1: the group's beams (coalesced loads, all issued before any is used)
#pragma unroll
        for (int k = 0; k < kGroup; ++k) {
            const int b = b0 + 64 * k + lane;
            const int bb = b < nbeams ? b : 0;
            qx[k] = bx[bb];
            qy[k] = by[bb];
        }
        // phase 2: cells and EDT gathers (out-of-bounds beams read cell 0 and are masked below)
#pragma unroll
        for (int k = 0; k < kGroup; ++k) {
            const int b = b0 + 64 * k + lane;
            const float px = qx[k] * g.ipix, py = qy[k] * g.ipix;
            const float rx = (px * ct) + (py * st);
            const float ry = (px * nst) + (py * ct);
            const float fx = round_half_away(rx + off_x);
            const float fy = round_half_away(ry + off_y);
            in[k] = (b < nbeams) & (fx > 0.0f) & (fy > 0.0f) & (fx < lim_x) & (fy < lim_y);
            h[k] = g.edt[in[k] ? (int)fy * g.ld + (int)fx : 0];
        }
        // phase 3: in-order slots
#pragma unroll
        for (int k = 0; k < kGroup; ++k) {
            const unsigned long long mask = __ballot(in[k]);
            const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
            if (in[k]) {
                s_hit[slot] = h[k];
                row[slot] = h[k];
            }
            base += __popcll(mask);
        }
    }
    __syncthreads();
    if (lane == 0) {
        float total = 0.0f;
#pragma unroll 8
        for (int j = 0; j < base; ++j) total = total + s_hit[j];
        out[c] = total;
        reinterpret_cast<int32_t*>(out)[kLatticeN + c] = base;
    }
}

// host_out / host_flag (optional): zero-copy result delivery — the same words are also written to pinned host
// memory mapped into the device, followed by a system-scope release of `seq` into *host_flag, so that the host
// can pick the result up by polling instead of paying a device-to-host copy and a stream synchronisation
// (a FastMatch call is latency-bound: ~40 us with copy + sync, ~20 us this way).
__global__ __launch_bounds__(256) void lattice_merge_kernel(const float* __restrict__ work, int nbeams,
                                                            float* __restrict__ out, float* __restrict__ persist,
                                                            float* __restrict__ host_out,
                                                            uint32_t* __restrict__ host_flag, uint32_t seq)
{
    __shared__ int s_cnt[kLatticeN];
    const int32_t* cnt = reinterpret_cast<const int32_t*>(out) + kLatticeN;
    if (threadIdx.x < kLatticeN) s_cnt[threadIdx.x] = cnt[threadIdx.x];
    __syncthreads();
    int maxc = 0;
    for (int c = 0; c < kLatticeN; ++c) maxc = s_cnt[c] > maxc ? s_cnt[c] : maxc;
    if (threadIdx.x == 0) reinterpret_cast<int32_t*>(out)[2 * kLatticeN] = maxc;
    float* merged = out + 2 * kLatticeN + 1;
    for (int j = threadIdx.x; j < maxc; j += 256) {
        int c = kLatticeN - 1;
        while (s_cnt[c] <= j) --c;   // terminates: some candidate has count == maxc > j
        const float v = work[(size_t)c * nbeams + j];
        merged[j] = v;
        if (persist) persist[j] = v;   // device-resident copy of the caller's persistent hit scratch
        if (host_out) host_out[2 * kLatticeN + 1 + j] = v;
    }
    if (host_out) {
        if (threadIdx.x < 2 * kLatticeN) host_out[threadIdx.x] = out[threadIdx.x];   // scores and counts (bit copies)
        if (threadIdx.x == 0) reinterpret_cast<int32_t*>(host_out)[2 * kLatticeN] = maxc;
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the merge of a chained pair (kernels.h: launch_lattice_pair)
__global__ __launch_bounds__(256) void lattice_merge_pair_kernel(const float* __restrict__ work1, const float* __restrict__ work2,
                                                                 int nbeams, float* __restrict__ out1, float* __restrict__ out2,
                                                                 float* __restrict__ persist, float* __restrict__ host_out1,
                                                                 float* __restrict__ host_out2, uint32_t* __restrict__ host_flag,
                                                                 uint32_t seq)
{
    __shared__ int s_c1[kLatticeN], s_c2[kLatticeN];
    if (threadIdx.x < kLatticeN) {
        s_c1[threadIdx.x] = reinterpret_cast<const int32_t*>(out1)[kLatticeN + threadIdx.x];
        s_c2[threadIdx.x] = reinterpret_cast<const int32_t*>(out2)[kLatticeN + threadIdx.x];
    }
    __syncthreads();
    int m1 = 0, m2 = 0;
    for (int c = 0; c < kLatticeN; ++c) {
        m1 = s_c1[c] > m1 ? s_c1[c] : m1;
        m2 = s_c2[c] > m2 ? s_c2[c] : m2;
    }
    if (threadIdx.x == 0) {
        reinterpret_cast<int32_t*>(out1)[2 * kLatticeN] = m1;
        reinterpret_cast<int32_t*>(out2)[2 * kLatticeN] = m2;
    }
    // entry j of the shared hit scratch after both sweeps: the LAST candidate of the second call with more than j in-bounds beams
    // wrote it last; where the second call has none, the first call's last such candidate did (main.c:515)
    const int top = m1 > m2 ? m1 : m2;
    if (persist)
        for (int j = threadIdx.x; j < top; j += 256) {
            const int* cnt = j < m2 ? s_c2 : s_c1;
            const float* work = j < m2 ? work2 : work1;
            int c = kLatticeN - 1;
            while (cnt[c] <= j) --c;   // terminates: that call has a candidate with count > j
            persist[j] = work[(size_t)c * nbeams + j];
        }
    if (threadIdx.x < 2 * kLatticeN) {
        host_out1[threadIdx.x] = out1[threadIdx.x];   // scores and counts (bit copies)
        host_out2[threadIdx.x] = out2[threadIdx.x];
    }
    if (threadIdx.x == 0) reinterpret_cast<int32_t*>(host_out2)[2 * kLatticeN] = m2;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

hipError_t launch_lattice_pair(hipStream_t stream, const ScoreGrid& g1, const ScoreGrid& g2, const float* bx, const float* by, int nbeams,
                               const int32_t* d_nbeams, const float* cand1, const float* pair_in, float* work1, float* work2, float* out1,
                               float* out2, float* persist, float* host_out1, float* host_out2, uint32_t* host_flag, uint32_t seq)
{
    const size_t lds = sizeof(float) * (size_t)(nbeams > 0 ? nbeams : 1);
    lattice_kernel<<<kLatticeN, 64, lds, stream>>>(g1, bx, by, nbeams, d_nbeams, cand1, work1, out1, nullptr, nullptr);
    lattice_kernel<<<kLatticeN, 64, lds, stream>>>(g2, bx, by, nbeams, d_nbeams, cand1, work2, out2, out1, pair_in);
    lattice_merge_pair_kernel<<<1, 256, 0, stream>>>(work1, work2, nbeams, out1, out2, persist, host_out1, host_out2, host_flag, seq);
    return hipGetLastError();
}

hipError_t launch_lattice(hipStream_t stream, const ScoreGrid& g, const float* bx, const float* by, int nbeams,
                          const int32_t* d_nbeams, const float* cand_xycs, float* work, float* out, float* persist,
                          float* host_out, uint32_t* host_flag, uint32_t seq)
{
    const size_t lds = sizeof(float) * (size_t)(nbeams > 0 ? nbeams : 1);
    lattice_kernel<<<kLatticeN, 64, lds, stream>>>(g, bx, by, nbeams, d_nbeams, cand_xycs, work, out, nullptr, nullptr);
    lattice_merge_kernel<<<1, 256, 0, stream>>>(work, nbeams, out, persist, host_out, host_flag, seq);
    return hipGetLastError();
}

namespace {
template <bool MOTION>
hipError_t launch_score_any(hipStream_t stream, const ScoreGrid& g, const float* bx, const float* by, int nbeams,
                            float* x, float* y, float* th_or_ct, const float* st_or_null, int nposes, float* score,
                            int32_t* count, const MotionIO& mio_in, const MotionParams& mpar, const EventPair* ev,
                            bool* rode = nullptr)
{
    if (rode) *rode = false;
    if (nposes <= 0) return hipSuccess;
    MotionIO mio = mio_in;
    // tuning knobs for measurements (pose counts below which the wave / quad lane mappings are used)
    static const int wave_max = getenv("SLAM_SCORE_WAVE_MAX") ? atoi(getenv("SLAM_SCORE_WAVE_MAX")) : kWaveMaxPoses;
    static const int quad_max = getenv("SLAM_SCORE_QUAD_MAX") ? atoi(getenv("SLAM_SCORE_QUAD_MAX")) : kQuadMaxPoses;
    if (nposes < wave_max) {   // one wavefront per pose
        const int blocks = (nposes + kScoreBlock / 64 - 1) / (kScoreBlock / 64);
        const size_t lds = sizeof(float) * (size_t)(kScoreBlock / 64) * (size_t)(nbeams > 0 ? nbeams : 1);
        if (ev) (void)hipEventRecord(ev->start, stream);
        if (st_or_null && !MOTION)
            score_poses_wave_kernel<true, MOTION><<<blocks, kScoreBlock, lds, stream>>>(
                g, bx, by, nbeams, x, y, th_or_ct, st_or_null, nposes, score, count, mio, mpar);
        else
            score_poses_wave_kernel<false, MOTION><<<blocks, kScoreBlock, lds, stream>>>(
                g, bx, by, nbeams, x, y, th_or_ct, st_or_null, nposes, score, count, mio, mpar);
        if (ev) (void)hipEventRecord(ev->stop, stream);
        return hipGetLastError();
    }
    const bool quad = nposes < quad_max;
    const long threads = quad ? 4L * nposes : nposes;
    const int score_blocks = (int)((threads + kScoreBlock - 1) / kScoreBlock);
    int blocks = score_blocks;
    if (MOTION && mio.rider.stamp) {   // + the rider's workgroups
        mio.rider.first_block = score_blocks;
        mio.rider.nblocks = free_list_blocks(mio.rider.npages);
        blocks += mio.rider.nblocks;
        if (rode) *rode = true;
    }
    const bool packed = g.packed != nullptr;   // the byte-per-cell copy of the grid (launch_edt_pack) + 1 KB of LDS for its table
    const size_t lds = sizeof(float2) * (size_t)(nbeams + (quad ? 4 * kQuadDepth : kLaneDepth)) + (packed ? 1024 : 0);
    if (ev) (void)hipEventRecord(ev->start, stream);
#define SLAM_LAUNCH_SCORE(CS, LPP, DEPTH, PK)                                                                              \
    score_poses_kernel<CS, LPP, DEPTH, MOTION, PK><<<blocks, kScoreBlock, lds, stream>>>(g, bx, by, nbeams, x, y, th_or_ct, \
                                                                                          st_or_null, nposes, score, count, \
                                                                                          mio, mpar)
#define SLAM_LAUNCH_SCORE2(CS, LPP, DEPTH)                 \
    do {                                                   \
        if (packed) SLAM_LAUNCH_SCORE(CS, LPP, DEPTH, true); \
        else SLAM_LAUNCH_SCORE(CS, LPP, DEPTH, false);       \
    } while (0)
    if (st_or_null && !MOTION) {
        if (quad) SLAM_LAUNCH_SCORE2(true, 4, kQuadDepth); else SLAM_LAUNCH_SCORE2(true, 1, kLaneDepth);
    } else {
        if (quad) SLAM_LAUNCH_SCORE2(false, 4, kQuadDepth); else SLAM_LAUNCH_SCORE2(false, 1, kLaneDepth);
    }
#undef SLAM_LAUNCH_SCORE2
#undef SLAM_LAUNCH_SCORE
    if (ev) (void)hipEventRecord(ev->stop, stream);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_score_poses(hipStream_t stream, const ScoreGrid& g, const float* bx, const float* by, int nbeams,
                              const float* x, const float* y, const float* th_or_ct, const float* st_or_null,
                              int nposes, float* score, int32_t* count, const EventPair* ev)
{
    // the non-MOTION instantiation never writes through x / y / th_or_ct
    return launch_score_any<false>(stream, g, bx, by, nbeams, const_cast<float*>(x), const_cast<float*>(y),
                                   const_cast<float*>(th_or_ct), st_or_null, nposes, score, count, MotionIO{},
                                   MotionParams{}, ev);
}

hipError_t launch_motion_score(hipStream_t stream, const ScoreGrid& g, const float* bx, const float* by, int nbeams,
                               const MotionIO& io, int nposes, int64_t first_id, const float dp[3], const float sigma[3],
                               uint64_t seed, uint32_t frame, float* score, int32_t* count, const EventPair* ev, bool* rode)
{
    return launch_score_any<true>(stream, g, bx, by, nbeams, io.x, io.y, io.th, nullptr, nposes, score, count, io,
                                  make_motion_params(first_id, dp, sigma, seed, frame), ev, rode);
}

// ---- the packed copy of a grid (kernels.h: ScoreGrid::packed).  Two passes: the grid's largest value (the cap, wherever a cell
// is farther than that from every occupied one), then code, table and the check that the table gives every cell back.
namespace {
__global__ __launch_bounds__(256) void edt_max_kernel(const float* __restrict__ edt, int ld, int rows, int cols, uint32_t* __restrict__ flag)
{
    uint32_t m = 0;   // the values are >= 0: their bit patterns order like the floats
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)rows * cols; i += (int64_t)gridDim.x * 256) {
        const float v = edt[(i / cols) * ld + i % cols];
        const uint32_t u = __float_as_uint(v);
        m = u > m ? u : m;   // a negative or NaN value ends up as a huge pattern: it fails the check of the second pass
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t t = (uint32_t)__shfl_xor((int)m, o);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(&flag[0], m);
}

__global__ __launch_bounds__(256) void edt_pack_kernel(const float* __restrict__ edt, int ld, int rows, int cols, int strip_bytes,
                                                       uint8_t* __restrict__ packed, float* __restrict__ table, uint32_t* __restrict__ flag)
{
    const float top = __uint_as_float(flag[0]);
    if (blockIdx.x == 0) table[threadIdx.x] = threadIdx.x == 255 ? top : sqrtf((float)threadIdx.x);
    const int rows8 = strip_bytes / 16;
    const int64_t cells = (int64_t)((cols + 15) / 16) * strip_bytes;   // every byte of the copy, padding included
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += (int64_t)gridDim.x * 256) {
        const int ix = (int)(i / strip_bytes) * 16 + (int)(i & 15), iy = (int)((i % strip_bytes) >> 4);
        unsigned code = 0;
        if (ix < cols && iy < rows && iy < rows8) {
            const float v = edt[(int64_t)iy * ld + ix];
            const float d2 = rintf(v * v);
            code = __float_as_uint(v) == __float_as_uint(top) ? 255u : (d2 >= 0.0f && d2 < 255.0f ? (unsigned)d2 : 254u);
            const float back = code == 255u ? top : sqrtf((float)code);
            bad = bad || __float_as_uint(back) != __float_as_uint(v);
        }
        packed[i] = (uint8_t)code;
    }
    if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) flag[1] = 1;
}
}  // namespace

size_t edt_packed_bytes(int rows, int cols) { return (size_t)((cols + 15) / 16) * 16 * (size_t)((rows + 7) / 8 * 8); }

hipError_t launch_edt_pack(hipStream_t stream, const float* edt, int ld, int rows, int cols, uint8_t* packed, float* table,
                           uint32_t* flag)
{
    hipError_t err = hipMemsetAsync(flag, 0, 8, stream);
    if (err != hipSuccess) return err;
    if (rows <= 0 || cols <= 0) return hipSuccess;
    const int64_t cells = (int64_t)rows * cols;
    const int blocks = (int)((cells + 255) / 256 < 2048 ? (cells + 255) / 256 : 2048);
    edt_max_kernel<<<blocks, 256, 0, stream>>>(edt, ld, rows, cols, flag);
    edt_pack_kernel<<<blocks, 256, 0, stream>>>(edt, ld, rows, cols, 16 * ((rows + 7) / 8 * 8), packed, table, flag);
    return hipGetLastError();
}

hipError_t launch_pose_hits(hipStream_t stream, const ScoreGrid& g, const float* bx, const float* by, int nbeams,
                            const float* pose_xycs, float* hits, int32_t* count)
{
    pose_hits_kernel<<<1, 64, 0, stream>>>(g, bx, by, nbeams, pose_xycs, hits, count);
    return hipGetLastError();
}

}  // namespace slam
