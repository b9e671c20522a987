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

#include "det_math.h"
#include "kernels.h"

namespace slam {

namespace {

constexpr int kScoreBlock = 256;

template <bool HAS_CS>
__global__ __launch_bounds__(kScoreBlock) void score_poses_kernel(ScoreGrid g, const float* __restrict__ bx,
                                                                   const float* __restrict__ by, int nbeams,
                                                                   const float* __restrict__ px,
                                                                   const float* __restrict__ py,
                                                                   const float* __restrict__ p2,
                                                                   const float* __restrict__ p3, int nposes,
                                                                   float* __restrict__ score,
                                                                   int32_t* __restrict__ count)
{
    extern __shared__ float2 s_beam[];
    for (int b = threadIdx.x; b < nbeams; b += kScoreBlock) s_beam[b] = make_float2(bx[b] * g.ipix, by[b] * g.ipix);
    __syncthreads();

    const int i = blockIdx.x * kScoreBlock + threadIdx.x;
    if (i >= nposes) return;

    float ct, st;
    if (HAS_CS) {
        ct = p2[i];
        st = p3[i];
    } else {
        det_sincosf(p2[i], st, ct);
    }
    const float nst = -st;
    const float off_x = (px[i] - g.min_x) * g.ipix;
    const float off_y = (py[i] - g.min_y) * g.ipix;
    // (int)roundf(v) + 1 > 1  <=>  roundf(v) > 0 ;  ... + 1 < n  <=>  roundf(v) < n - 1  (integers in float)
    const float lim_x = (float)(g.cols - 1);
    const float lim_y = (float)(g.rows - 1);
    const float* __restrict__ edt = g.edt;
    const int ld = g.ld;

    float total = 0.0f;
    int n_in = 0;
#pragma unroll 8
    for (int b = 0; b < nbeams; ++b) {
        const float2 q = s_beam[b];
        const float rx = (q.x * ct) + (q.y * st);
        const float ry = (q.x * nst) + (q.y * ct);
        const float fx = round_half_away(rx + off_x);
        const float fy = round_half_away(ry + off_y);
        const bool in = (fx > 0.0f) & (fy > 0.0f) & (fx < lim_x) & (fy < lim_y);
        const int idx = in ? (int)fy * ld + (int)fx : 0;
        const float h = edt[idx];
        total = total + (in ? h : 0.0f);   // adding +0 leaves the running sum's bits unchanged
        n_in += in ? 1 : 0;
    }
    score[i] = total;
    count[i] = n_in;
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
                                                     const float* __restrict__ by, int nbeams,
                                                     const float* __restrict__ cand, float* __restrict__ work,
                                                     float* __restrict__ out)
{
    extern __shared__ float s_hit[];
    const int c = blockIdx.x, lane = threadIdx.x;
    const float x = cand[c], y = cand[kLatticeN + c], ct = cand[2 * kLatticeN + c], st = cand[3 * kLatticeN + c];
    const float nst = -st;
    const float off_x = (x - g.min_x) * g.ipix;
    const float off_y = (y - g.min_y) * g.ipix;
    const float lim_x = (float)(g.cols - 1);
    const float lim_y = (float)(g.rows - 1);
    float* __restrict__ row = work + (size_t)c * nbeams;
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
        const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
        if (in) {
            s_hit[slot] = h;
            row[slot] = h;
        }
        base += __popcll(mask);
    }
    __syncthreads();
    if (lane == 0) {
        float total = 0.0f;
        for (int j = 0; j < base; ++j) total = total + s_hit[j];
        out[c] = total;
        reinterpret_cast<int32_t*>(out)[kLatticeN + c] = base;
    }
}

__global__ __launch_bounds__(256) void lattice_merge_kernel(const float* __restrict__ work, int nbeams,
                                                            float* __restrict__ out)
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
        merged[j] = work[(size_t)c * nbeams + j];
    }
}

}  // namespace

hipError_t launch_lattice(hipStream_t stream, const ScoreGrid& g, const float* bx, const float* by, int nbeams,
                          const float* cand_xycs, float* work, float* out)
{
    const size_t lds = sizeof(float) * (size_t)(nbeams > 0 ? nbeams : 1);
    lattice_kernel<<<kLatticeN, 64, lds, stream>>>(g, bx, by, nbeams, cand_xycs, work, out);
    lattice_merge_kernel<<<1, 256, 0, stream>>>(work, nbeams, out);
    return hipGetLastError();
}

hipError_t launch_score_poses(hipStream_t stream, const ScoreGrid& g, const float* bx, const float* by, int nbeams,
                              const float* x, const float* y, const float* th_or_ct, const float* st_or_null,
                              int nposes, float* score, int32_t* count, const EventPair* ev)
{
    if (nposes <= 0) return hipSuccess;
    const int blocks = (nposes + kScoreBlock - 1) / kScoreBlock;
    const size_t lds = sizeof(float2) * (size_t)(nbeams > 0 ? nbeams : 1);
    if (ev) (void)hipEventRecord(ev->start, stream);
    if (st_or_null)
        score_poses_kernel<true><<<blocks, kScoreBlock, lds, stream>>>(g, bx, by, nbeams, x, y, th_or_ct, st_or_null,
                                                                        nposes, score, count);
    else
        score_poses_kernel<false><<<blocks, kScoreBlock, lds, stream>>>(g, bx, by, nbeams, x, y, th_or_ct, nullptr,
                                                                         nposes, score, count);
    if (ev) (void)hipEventRecord(ev->stop, stream);
    return hipGetLastError();
}

hipError_t launch_pose_hits(hipStream_t stream, const ScoreGrid& g, const float* bx, const float* by, int nbeams,
                            const float* pose_xycs, float* hits, int32_t* count)
{
    pose_hits_kernel<<<1, 64, 0, stream>>>(g, bx, by, nbeams, pose_xycs, hits, count);
    return hipGetLastError();
}

}  // namespace slam
