// pf_kernels.hip — particle-filter stages around the scan matcher.
//
// None of these stages exists in the reference: the specification is DESIGN.md
// + oracle/slam_oracle_pf.c, and these kernels match that specification bit for bit.  The only
// reference anchor is the zero-noise motion step = the constant-velocity predict of
// Subsystem_1/main.c:875-898.
//
// Data layout (HBM): particles are SoA float arrays; the per-particle landmark maps are five
// planes [L][ld] (mu_x, mu_y, P_xx, P_xy, P_yy) with the particle index fastest, so a wavefront
// reads/writes 64 consecutive particles of one landmark = 256 contiguous bytes per plane.
// All kernels are HBM-streaming or latency-bound integer work; there is no GEMM shape here
// (the largest matrix is 2x2), hence no MFMA.

#include "det_math.h"
#include "kernels.h"

namespace slam {

namespace {

constexpr int kBlock = 256;

// ------------------------------------------------------------------ A9: motion sample
__global__ __launch_bounds__(kBlock) void motion_sample_kernel(const float* __restrict__ sx,
                                                               const float* __restrict__ sy,
                                                               const float* __restrict__ sth,
                                                               const int32_t* __restrict__ anc, float* __restrict__ x,
                                                               float* __restrict__ y, float* __restrict__ th, int n,
                                                               MotionParams mp)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int j = anc ? anc[i] : i;
    float ox, oy, ot;
    motion_sample_one(mp, (uint64_t)i, sx[j], sy[j], sth[j], ox, oy, ot);
    x[i] = ox;
    y[i] = oy;
    th[i] = ot;
}

// ------------------------------------------------------------------ A10: 2x2 EKF per (particle, landmark)
// grid = (particle tiles, observation chunks).  A thread owns one particle and walks one chunk of
// EKF_OBS_CHUNK observations; consecutive lanes = consecutive particles, so each of the 5 loads and 5
// stores per landmark is a 256-byte coalesced wave access.  The observation list (id, zx, zy) is
// wave-uniform and read through the scalar path.
__global__ __launch_bounds__(kBlock) void ekf_update_kernel(EkfArgs a)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n) return;
    const int chunk = blockIdx.y;
    const int k0 = chunk * EKF_OBS_CHUNK;
    const int k1 = min(k0 + EKF_OBS_CHUNK, a.nobs);

    float st, ct;
    det_sincosf(a.th[i], st, ct);
    const float px = a.x[i], py = a.y[i];
    const int src = a.anc ? a.anc[i] : i;
    const float q = a.meas_var;
    const int64_t ps = a.plane_stride;
    const float* __restrict__ in = a.map_in;
    float* __restrict__ out = a.map_out;

    float part = 0.0f;
#pragma unroll 4
    for (int k = k0; k < k1; ++k) {
        const int64_t row = (int64_t)a.obs_id[k] * a.ld;
        const int64_t ri = row + src, wi = row + i;
        const float mx = in[ri], my = in[ps + ri], pxx = in[2 * ps + ri], pxy = in[3 * ps + ri],
                    pyy = in[4 * ps + ri];
        const float zx = a.obs_zx[k], zy = a.obs_zy[k];
        float omx, omy, oxx, oxy, oyy;
        if (pxx < 0.0f) {   // first sighting
            omx = px + (ct * zx + st * zy);
            omy = py + (ct * zy - st * zx);
            oxx = q;
            oxy = 0.0f;
            oyy = q;
        } else {
            const float dx = mx - px, dy = my - py;
            const float vx = zx - (ct * dx - st * dy);
            const float vy = zy - (st * dx + ct * dy);
            const float a00 = ct * pxx - st * pxy, a01 = ct * pxy - st * pyy;
            const float a10 = st * pxx + ct * pxy, a11 = st * pxy + ct * pyy;
            const float s00 = (a00 * ct - a01 * st) + q;
            const float s01 = a00 * st + a01 * ct;
            const float s11 = (a10 * st + a11 * ct) + q;
            const float det = s00 * s11 - s01 * s01;
            const float idet = 1.0f / det;
            const float i00 = s11 * idet, i01 = -s01 * idet, i11 = s00 * idet;
            const float k00 = a00 * i00 + a10 * i01, k01 = a00 * i01 + a10 * i11;
            const float k10 = a01 * i00 + a11 * i01, k11 = a01 * i01 + a11 * i11;
            omx = mx + (k00 * vx + k01 * vy);
            omy = my + (k10 * vx + k11 * vy);
            oxx = pxx - (k00 * a00 + k01 * a10);
            oxy = pxy - (k00 * a01 + k01 * a11);
            oyy = pyy - (k10 * a01 + k11 * a11);
            const float maha = vx * (i00 * vx + i01 * vy) + vy * (i01 * vx + i11 * vy);
            part = ((part - 0.5f * maha) - 0.5f * det_logf(det)) - 1.8378770664f;
        }
        out[wi] = omx;
        out[ps + wi] = omy;
        out[2 * ps + wi] = oxx;
        out[3 * ps + wi] = oxy;
        out[4 * ps + wi] = oyy;
    }
    if (gridDim.y == 1 && a.loglik)
        a.loglik[i] = 0.0f + part;
    else
        a.ll_part[(int64_t)chunk * a.n + i] = part;
}

__global__ __launch_bounds__(kBlock) void ekf_loglik_finalize_kernel(const float* __restrict__ part, int nchunks, int n,
                                                                     float* __restrict__ loglik)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float total = 0.0f;
    for (int c = 0; c < nchunks; ++c) total = total + part[(int64_t)c * n + i];
    loglik[i] = total;
}

// landmarks without an observation: gathered copy in -> out (only for out-of-place updates)
__global__ __launch_bounds__(kBlock) void map_copy_through_kernel(EkfArgs a)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n) return;
    const int src = a.anc ? a.anc[i] : i;
    const int64_t ps = a.plane_stride;
    for (int u = blockIdx.y; u < a.nunobs; u += gridDim.y) {
        const int64_t row = (int64_t)a.unobs_id[u] * a.ld;
#pragma unroll
        for (int p = 0; p < 5; ++p) a.map_out[p * ps + row + i] = a.map_in[p * ps + row + src];
    }
}

// ------------------------------------------------------------------ A11: weights
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ll_part/nchunks: when the EKF left its per-chunk partial sums (nchunks > 1) they are added up here in
// chunk order — the specified summation order — instead of in a separate finalize pass.
__global__ __launch_bounds__(kBlock) void logweight_kernel(const float* __restrict__ score,
                                                           const float* __restrict__ loglik,
                                                           const float* __restrict__ ll_part, int nchunks, float gain,
                                                           int n, float* __restrict__ logw,
                                                           float* __restrict__ block_max)
{
    __shared__ float s_max[kBlock / 64];
    float m = -INFINITY;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        float ll = loglik ? loglik[i] : 0.0f;
        if (ll_part) {
            ll = 0.0f;
            for (int c = 0; c < nchunks; ++c) ll = ll + ll_part[(int64_t)c * n + i];
        }
        const float sc = score ? score[i] * gain : 0.0f;
        const float lw = ll - sc;
        logw[i] = lw;
        m = lw > m ? lw : m;
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) m = fmaxf(m, s_max[w]);
        block_max[blockIdx.x] = m;
    }
}

__global__ __launch_bounds__(kBlock) void max_finalize_kernel(const float* __restrict__ block_max, int nblocks,
                                                              float* __restrict__ d_max)
{
    __shared__ float s_max[kBlock / 64];
    float m = -INFINITY;
    for (int i = threadIdx.x; i < nblocks; i += kBlock) m = fmaxf(m, block_max[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) m = fmaxf(m, s_max[w]);
        *d_max = m;
    }
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, o), hi = __shfl_xor((uint32_t)(v >> 32), o);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

__global__ __launch_bounds__(kBlock) void quantise_weights_kernel(const float* __restrict__ logw,
                                                                  const float* __restrict__ d_max, int n,
                                                                  uint64_t* __restrict__ wq,
                                                                  unsigned long long* __restrict__ d_sum)
{
    __shared__ uint64_t s_sum[kBlock / 64];
    const float m = *d_max;
    uint64_t acc = 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float w = det_expf(logw[i] - m);
        const uint64_t q = (uint64_t)(w * 4294967296.0f);
        wq[i] = q;
        acc += q;
    }
    acc = wave_sum_u64(acc);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) acc += s_sum[w];
        atomicAdd(d_sum, (unsigned long long)acc);   // integer: exact and order-independent
    }
}

// ------------------------------------------------------------------ A12: integer CDF, comb, ancestors
constexpr int kScanItems = 8;                       // elements per thread
constexpr int kScanTile = kBlock * kScanItems;      // 2048 elements per workgroup

__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int d)
{
    const uint32_t lo = __shfl_up((uint32_t)v, d), hi = __shfl_up((uint32_t)(v >> 32), d);
    return ((uint64_t)hi << 32) | lo;
}

// inclusive scan of one value per thread over the workgroup: wavefront scan + carry through LDS
__device__ __forceinline__ uint64_t block_inclusive_scan(uint64_t v, uint64_t* s_wave /*[kBlock/64]*/, uint64_t& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t up = shfl_up_u64(v, d);
        if (lane >= d) v += up;
    }
    if (lane == 63) s_wave[wave] = v;
    __syncthreads();
    uint64_t carry = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
        if (w < wave) carry += s_wave[w];
        tot += s_wave[w];
    }
    total = tot;
    __syncthreads();
    return v + carry;
}

__global__ __launch_bounds__(kBlock) void scan_tiles_kernel(const uint64_t* __restrict__ in, int n,
                                                            uint64_t* __restrict__ out,
                                                            uint64_t* __restrict__ tile_total)
{
    __shared__ uint64_t s_wave[kBlock / 64];
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint64_t v[kScanItems];
    uint64_t run = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int i = base + k;
        run += i < n ? in[i] : 0;
        v[k] = run;
    }
    uint64_t total;
    const uint64_t incl = block_inclusive_scan(run, s_wave, total);
    const uint64_t excl = incl - run;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int i = base + k;
        if (i < n) out[i] = v[k] + excl;
    }
    if (threadIdx.x == 0) tile_total[blockIdx.x] = total;
}

// exclusive scan of the tile totals, in place, by ONE workgroup (<= a few thousand tiles)
__global__ __launch_bounds__(kBlock) void scan_totals_kernel(uint64_t* __restrict__ tile_total, int ntiles)
{
    __shared__ uint64_t s_wave[kBlock / 64];
    uint64_t carry = 0;
    for (int t0 = 0; t0 < ntiles; t0 += kBlock) {
        const int t = t0 + threadIdx.x;
        const uint64_t v = t < ntiles ? tile_total[t] : 0;
        uint64_t total;
        const uint64_t incl = block_inclusive_scan(v, s_wave, total);
        if (t < ntiles) tile_total[t] = carry + incl - v;
        carry += total;
    }
}

__global__ __launch_bounds__(kBlock) void add_tile_offsets_kernel(uint64_t* __restrict__ out, int n,
                                                                  const uint64_t* __restrict__ tile_excl)
{
    const uint64_t off = tile_excl[blockIdx.x];
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int i = base + k;
        if (i < n) out[i] += off;
    }
}

// ---- fused frame-loop form: weights are quantised and scanned in one pass, never stored
// max over an array of block maxima, by the whole workgroup (every workgroup repeats it: <= 2048 floats)
__device__ __forceinline__ float block_max_of(const float* __restrict__ v, int count, float* s_red /*[kBlock/64]*/)
{
    float m = -INFINITY;
    for (int i = threadIdx.x; i < count; i += kBlock) m = fmaxf(m, v[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    float r = s_red[0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) r = fmaxf(r, s_red[w]);
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(kBlock) void quantise_scan_kernel(const float* __restrict__ logw,
                                                               const float* __restrict__ d_max,
                                                               const float* __restrict__ block_max, int nblock_max,
                                                               int n, uint64_t* __restrict__ cdf_local,
                                                               uint64_t* __restrict__ tile_total)
{
    __shared__ uint64_t s_wave[kBlock / 64];
    __shared__ float s_red[kBlock / 64];
    const float m = d_max ? *d_max : block_max_of(block_max, nblock_max, s_red);
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint64_t v[kScanItems];
    uint64_t run = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int i = base + k;
        uint64_t q = 0;
        if (i < n) q = (uint64_t)(det_expf(logw[i] - m) * 4294967296.0f);
        run += q;
        v[k] = run;
    }
    uint64_t total;
    const uint64_t incl = block_inclusive_scan(run, s_wave, total);
    const uint64_t excl = incl - run;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int i = base + k;
        if (i < n) cdf_local[i] = v[k] + excl;   // inclusive, local to this 2048-element tile
    }
    if (threadIdx.x == 0) tile_total[blockIdx.x] = total;
}

__device__ __forceinline__ uint64_t block_sum_u64(uint64_t v, uint64_t* s_red /*[kBlock/64]*/)
{
    v = wave_sum_u64(v);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    uint64_t r = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) r += s_red[w];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(kBlock) void sum_tiles_kernel(const uint64_t* __restrict__ tile_total, int ntiles,
                                                           uint64_t* __restrict__ d_sum)
{
    __shared__ uint64_t s_red[kBlock / 64];
    uint64_t acc = 0;
    for (int t = threadIdx.x; t < ntiles; t += kBlock) acc += tile_total[t];
    acc = block_sum_u64(acc, s_red);
    if (threadIdx.x == 0) *d_sum = acc;
}

// floor((hi:lo) / d) for hi < d < 2^63 (so the quotient fits 64 bits): restoring long division
__device__ __forceinline__ uint64_t div128by64(uint64_t hi, uint64_t lo, uint64_t d)
{
    uint64_t rem = hi, q = 0;
#pragma unroll 8
    for (int b = 63; b >= 0; --b) {
        rem = (rem << 1) | ((lo >> b) & 1ull);
        if (rem >= d) {
            rem -= d;
            q |= 1ull << b;
        }
    }
    return q;
}

__global__ __launch_bounds__(kBlock) void offspring_offsets_kernel(const uint64_t* __restrict__ cdf, int n,
                                                                   const uint64_t* __restrict__ d_base,
                                                                   const uint64_t* __restrict__ d_total,
                                                                   uint32_t key0, uint32_t key1, uint32_t frame,
                                                                   uint64_t n_total, int32_t* __restrict__ first)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint64_t base = d_base ? *d_base : 0ull;
    const uint64_t total = *d_total;
    if (total == 0 || (total >> 63)) {   // impossible with finite log-weights (the best particle has wq = 2^32);
        first[i] = 0;                    // stay memory-safe anyway: no division, every slot gets the last particle
        return;
    }
    // comb offset u in [0,total): Philox counter (0,0,frame,1), high 64 bits of r64*total (wave-uniform)
    const u32x4 r = philox4x32_10(0u, 0u, frame, 1u /* resample stream */, key0, key1);
    const uint64_t comb_u = __umul64hi((uint64_t)r.v[0] | ((uint64_t)r.v[1] << 32), total);
    const uint64_t c_excl = base + (i ? cdf[i - 1] : 0ull);
    // X = c_excl * n_total as 128 bits
    uint64_t lo = c_excl * n_total, hi = __umul64hi(c_excl, n_total);
    int32_t f = 0;
    if (hi != 0 || lo > comb_u) {
        // (X - u - 1) / total + 1
        const uint64_t sub = comb_u + 1ull;   // comb_u < total < 2^63: no overflow
        hi -= lo < sub ? 1ull : 0ull;
        lo -= sub;
        f = (int32_t)(div128by64(hi, lo, total) + 1ull);
    }
    first[i] = f;
}

// comb_first(): shared by the staged and the fused offsets kernels
__device__ __forceinline__ int32_t comb_first(uint64_t c_excl, uint64_t total, uint64_t n_total, uint32_t key0,
                                              uint32_t key1, uint32_t frame)
{
    if (total == 0 || (total >> 63)) return 0;   // see offspring_offsets_kernel
    const u32x4 r = philox4x32_10(0u, 0u, frame, 1u /* resample stream */, key0, key1);
    const uint64_t comb_u = __umul64hi((uint64_t)r.v[0] | ((uint64_t)r.v[1] << 32), total);
    uint64_t lo = c_excl * n_total, hi = __umul64hi(c_excl, n_total);
    if (hi == 0 && lo <= comb_u) return 0;
    const uint64_t sub = comb_u + 1ull;
    hi -= lo < sub ? 1ull : 0ull;
    lo -= sub;
    return (int32_t)(div128by64(hi, lo, total) + 1ull);
}

// fused form: CDF = base + (sum of earlier tiles) + tile-local scan; every workgroup re-derives its tile's
// offset (and, on a single GPU, the grand total) from the <= n/2048 tile totals instead of a separate pass
__global__ __launch_bounds__(kBlock) void offspring_from_scan_kernel(const uint64_t* __restrict__ cdf_local,
                                                                     const uint64_t* __restrict__ tile_total,
                                                                     int ntiles, int n,
                                                                     const uint64_t* __restrict__ d_base,
                                                                     const uint64_t* __restrict__ d_total,
                                                                     const uint64_t* __restrict__ d_shard_totals,
                                                                     int rank, int world, uint32_t key0,
                                                                     uint32_t key1, uint32_t frame, uint64_t n_total,
                                                                     int32_t* __restrict__ first)
{
    __shared__ uint64_t s_red[kBlock / 64];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int tile = (blockIdx.x * kBlock) / kScanTile;   // kScanTile is a multiple of kBlock
    uint64_t before = 0, all = 0;
    for (int t = threadIdx.x; t < ntiles; t += kBlock) {
        const uint64_t v = tile_total[t];
        all += v;
        before += t < tile ? v : 0ull;
    }
    before = block_sum_u64(before, s_red);
    uint64_t total, shard_base = d_base ? *d_base : 0ull;
    if (d_shard_totals) {   // several GPUs: the all-gathered shard totals give both the base and the grand total
        total = 0;
        shard_base = 0;
        for (int q = 0; q < world; ++q) {
            const uint64_t v = d_shard_totals[q];
            total += v;
            shard_base += q < rank ? v : 0ull;
        }
    } else {
        total = d_total ? *d_total : block_sum_u64(all, s_red);
    }
    if (i >= n) return;
    const uint64_t base = shard_base + before;
    const uint64_t c_excl = base + ((i % kScanTile) ? cdf_local[i - 1] : 0ull);
    first[i] = comb_first(c_excl, total, n_total, key0, key1, frame);
}

__global__ __launch_bounds__(kBlock) void ancestors_kernel(const int32_t* __restrict__ first_all, int64_t n_total,
                                                           int64_t slot0, int nslots, int32_t* __restrict__ anc)
{
    const int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= nslots) return;
    const int64_t j = slot0 + s;
    int64_t lo = 0, hi = n_total;   // first index whose first slot is > j
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)first_all[mid] <= j) lo = mid + 1; else hi = mid;
    }
    anc[s] = (int32_t)(lo - 1);
}

// ------------------------------------------------------------------ multi-GPU resample: sharded gather index + migration
// Slots of rank r are [r*n, (r+1)*n).  The particles of rank s fill the slot range [A_s, B_s) with
// A_s = first_all[s*n], B_s = A_(s+1) (B of the last rank = n_total) because `first` is non-decreasing.
// So what rank r receives from rank s is ONE contiguous run of its slots, and everything below follows
// from the world+1 boundary values — no host-computed plan is needed for the index kernel.
__device__ __forceinline__ int64_t first_or_total(const int32_t* __restrict__ first_all, int64_t idx, int64_t n_total)
{
    return idx < n_total ? (int64_t)first_all[idx] : n_total;
}

__device__ __forceinline__ int64_t last_with_first_le(const int32_t* __restrict__ first_all, int64_t n_total, int64_t j)
{
    int64_t lo = 0, hi = n_total;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)first_all[mid] <= j) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}

// src[jl]: where slot r*n + jl finds its ancestor — a local particle index, or n + (position in the staging
// tail, which holds the runs received from ranks 0..world-1 in rank order, each run in slot order).
__global__ __launch_bounds__(kBlock) void ancestors_sharded_kernel(const int32_t* __restrict__ first_all,
                                                                   int64_t n_total, int n, int rank, int world,
                                                                   int32_t* __restrict__ src)
{
    __shared__ int64_t s_lo[kMaxRanks];    // first of my slots served by rank s
    __shared__ int32_t s_off[kMaxRanks];   // staging offset of rank s's run
    if (threadIdx.x == 0) {
        const int64_t my_lo = (int64_t)rank * n, my_hi = my_lo + n;
        int32_t off = 0;
        for (int q = 0; q < world; ++q) {
            const int64_t a = first_or_total(first_all, (int64_t)q * n, n_total);
            const int64_t b = first_or_total(first_all, (int64_t)(q + 1) * n, n_total);
            const int64_t lo = a > my_lo ? a : my_lo, hi = b < my_hi ? b : my_hi;
            s_lo[q] = lo;
            s_off[q] = off;
            if (q != rank && hi > lo) off += (int32_t)(hi - lo);
        }
    }
    __syncthreads();
    const int jl = blockIdx.x * kBlock + threadIdx.x;
    if (jl >= n) return;
    const int64_t j = (int64_t)rank * n + jl;
    const int64_t g = last_with_first_le(first_all, n_total, j);
    const int owner = (int)(g / n);
    src[jl] = owner == rank ? (int32_t)(g - (int64_t)rank * n) : n + s_off[owner] + (int32_t)(j - s_lo[owner]);
}

// Pack what the other ranks need from me into one buffer: block d (for rank d) is [3 + 5L][cnt_d] floats —
// rows x, y, theta, then the 5 map planes landmark by landmark — for the cnt_d consecutive slots starting
// at lo_d whose ancestors are my particles.  One launch for every destination.
__global__ __launch_bounds__(kBlock) void migrate_pack_kernel(const int32_t* __restrict__ first_all, int64_t n_total,
                                                              int n, int rank, MigratePlan plan,
                                                              const float* __restrict__ pose, int64_t pose_ld,
                                                              const float* __restrict__ map, int64_t plane_stride,
                                                              int ld, int nlandmarks, float* __restrict__ out)
{
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= plan.off[plan.world]) return;
    int d = 0;
    while (p >= plan.off[d + 1]) ++d;
    const int q = p - plan.off[d], cnt = plan.off[d + 1] - plan.off[d];
    const int64_t j = plan.lo[d] + q;
    const int loc = (int)(last_with_first_le(first_all, n_total, j) - (int64_t)rank * n);
    const int rows = 3 + 5 * nlandmarks;
    float* __restrict__ blk = out + (int64_t)rows * plan.off[d] + q;
    for (int k = blockIdx.y; k < rows; k += gridDim.y) {
        float v;
        if (k < 3) {
            v = pose[k * pose_ld + loc];
        } else {
            const int m = k - 3, pl = m / nlandmarks, l = m - pl * nlandmarks;
            v = map[pl * plane_stride + (int64_t)l * ld + loc];
        }
        blk[(int64_t)k * cnt] = v;
    }
}

// Unpack the received blocks into the staging tail behind the n local particles (position = running index
// over sources in rank order, matching ancestors_sharded_kernel).
__global__ __launch_bounds__(kBlock) void migrate_unpack_kernel(const float* __restrict__ in, MigratePlan plan, int n,
                                                                float* __restrict__ pose, int64_t pose_ld,
                                                                float* __restrict__ map, int64_t plane_stride, int ld,
                                                                int nlandmarks)
{
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= plan.off[plan.world]) return;
    int s = 0;
    while (p >= plan.off[s + 1]) ++s;
    const int q = p - plan.off[s], cnt = plan.off[s + 1] - plan.off[s];
    const int rows = 3 + 5 * nlandmarks;
    const float* __restrict__ blk = in + (int64_t)rows * plan.off[s] + q;
    for (int k = blockIdx.y; k < rows; k += gridDim.y) {
        const float v = blk[(int64_t)k * cnt];
        if (k < 3) {
            pose[k * pose_ld + n + p] = v;
        } else {
            const int m = k - 3, pl = m / nlandmarks, l = m - pl * nlandmarks;
            map[pl * plane_stride + (int64_t)l * ld + n + p] = v;
        }
    }
}

__global__ __launch_bounds__(kBlock) void gather_f32_kernel(const float* __restrict__ src,
                                                            const int32_t* __restrict__ idx, int n,
                                                            float* __restrict__ dst)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

__global__ __launch_bounds__(kBlock) void gather_map_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int64_t in_stride, int64_t out_stride, int ld_in,
                                                            int ld_out, int nlandmarks,
                                                            const int32_t* __restrict__ idx, int n)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int src = idx[i];
    for (int l = blockIdx.y; l < nlandmarks; l += gridDim.y) {
#pragma unroll
        for (int p = 0; p < 5; ++p)
            out[p * out_stride + (int64_t)l * ld_out + i] = in[p * in_stride + (int64_t)l * ld_in + src];
    }
}

// index of the largest value, lowest index on ties (the heaviest particle); one workgroup
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ v, int n, int32_t* __restrict__ idx_out,
                                                      float* __restrict__ val_out)
{
    __shared__ float s_v[16];
    __shared__ int s_i[16];
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float x = v[i];
        if (x > bv || (x == bv && i < bi)) { bv = x; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = bv; s_i[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) { bv = s_v[w]; bi = s_i[w]; }
        *idx_out = bi == 0x7fffffff ? 0 : bi;
        *val_out = bv;
    }
}

inline int blocks_for(int n) { return (n + kBlock - 1) / kBlock; }

}  // namespace

hipError_t launch_motion_sample(hipStream_t stream, const float* sx, const float* sy, const float* sth,
                                const int32_t* anc, float* x, float* y, float* th, int n, int64_t first_id,
                                const float dp[3], const float sigma[3], uint64_t seed, uint32_t frame)
{
    if (n <= 0) return hipSuccess;
    const MotionParams mp = make_motion_params(first_id, dp, sigma, seed, frame);
    motion_sample_kernel<<<blocks_for(n), kBlock, 0, stream>>>(sx, sy, sth, anc, x, y, th, n, mp);
    return hipGetLastError();
}

hipError_t launch_ekf_update(hipStream_t stream, const EkfArgs& a, const EventPair* ev)
{
    if (a.n <= 0) return hipSuccess;
    const int nchunks = (a.nobs + EKF_OBS_CHUNK - 1) / EKF_OBS_CHUNK;
    if (a.map_in != a.map_out && a.nunobs > 0) {
        const int gy = a.nunobs < 1024 ? a.nunobs : 1024;
        map_copy_through_kernel<<<dim3(blocks_for(a.n), gy), kBlock, 0, stream>>>(a);
    }
    if (nchunks == 0) {
        if (!a.loglik) return hipGetLastError();
        hipError_t err = hipMemsetAsync(a.loglik, 0, sizeof(float) * (size_t)a.n, stream);
        return err != hipSuccess ? err : hipGetLastError();
    }
    if (ev) (void)hipEventRecord(ev->start, stream);
    ekf_update_kernel<<<dim3(blocks_for(a.n), nchunks), kBlock, 0, stream>>>(a);
    if (ev) (void)hipEventRecord(ev->stop, stream);
    if (nchunks > 1 && a.loglik)   // loglik == nullptr: the partials are consumed by launch_logweight instead
        ekf_loglik_finalize_kernel<<<blocks_for(a.n), kBlock, 0, stream>>>(a.ll_part, nchunks, a.n, a.loglik);
    return hipGetLastError();
}

static int capped_blocks(int n) { const int b = blocks_for(n); return b < 2048 ? b : 2048; }
int logweight_scratch_elems(int n) { return capped_blocks(n > 0 ? n : 1); }

hipError_t launch_logweight(hipStream_t stream, const float* score, const float* loglik, const float* ll_part,
                            int nchunks, float gain, int n, float* logw, float* block_max_scratch, float* d_max)
{
    if (n <= 0) return hipSuccess;
    const int nb = capped_blocks(n);
    logweight_kernel<<<nb, kBlock, 0, stream>>>(score, loglik, ll_part, nchunks, gain, n, logw, block_max_scratch);
    if (d_max) max_finalize_kernel<<<1, kBlock, 0, stream>>>(block_max_scratch, nb, d_max);
    return hipGetLastError();
}

hipError_t launch_quantise_weights(hipStream_t stream, const float* logw, const float* d_max, int n, uint64_t* wq,
                                   uint64_t* d_sum)
{
    hipError_t err = hipMemsetAsync(d_sum, 0, sizeof(uint64_t), stream);
    if (err != hipSuccess) return err;
    if (n <= 0) return hipSuccess;
    quantise_weights_kernel<<<capped_blocks(n), kBlock, 0, stream>>>(logw, d_max, n, wq,
                                                                     reinterpret_cast<unsigned long long*>(d_sum));
    return hipGetLastError();
}

int prefix_sum_scratch_elems(int n) { return (n + kScanTile - 1) / kScanTile + 1; }

hipError_t launch_prefix_sum(hipStream_t stream, const uint64_t* in, int n, uint64_t* out, uint64_t* block_scratch)
{
    if (n <= 0) return hipSuccess;
    const int ntiles = (n + kScanTile - 1) / kScanTile;
    scan_tiles_kernel<<<ntiles, kBlock, 0, stream>>>(in, n, out, block_scratch);
    if (ntiles > 1) {
        scan_totals_kernel<<<1, kBlock, 0, stream>>>(block_scratch, ntiles);
        add_tile_offsets_kernel<<<ntiles, kBlock, 0, stream>>>(out, n, block_scratch);
    }
    return hipGetLastError();
}

hipError_t launch_quantise_scan(hipStream_t stream, const float* logw, const float* d_max, const float* block_max,
                                int nblock_max, int n, uint64_t* cdf_local, uint64_t* tile_total, uint64_t* d_sum)
{
    if (n <= 0) return hipSuccess;
    const int ntiles = (n + kScanTile - 1) / kScanTile;
    quantise_scan_kernel<<<ntiles, kBlock, 0, stream>>>(logw, d_max, block_max, nblock_max, n, cdf_local, tile_total);
    if (d_sum) sum_tiles_kernel<<<1, kBlock, 0, stream>>>(tile_total, ntiles, d_sum);
    return hipGetLastError();
}

hipError_t launch_offspring_from_scan(hipStream_t stream, const uint64_t* cdf_local, const uint64_t* tile_total, int n,
                                      const uint64_t* d_base, const uint64_t* d_total, const uint64_t* d_shard_totals,
                                      int rank, int world, uint64_t seed, uint32_t frame, int64_t n_total,
                                      int32_t* first)
{
    if (n <= 0) return hipSuccess;
    const int ntiles = (n + kScanTile - 1) / kScanTile;
    offspring_from_scan_kernel<<<blocks_for(n), kBlock, 0, stream>>>(cdf_local, tile_total, ntiles, n, d_base, d_total,
                                                                     d_shard_totals, rank, world, (uint32_t)seed,
                                                                     (uint32_t)(seed >> 32), frame, (uint64_t)n_total,
                                                                     first);
    return hipGetLastError();
}

hipError_t launch_offspring_offsets(hipStream_t stream, const uint64_t* cdf, int n, const uint64_t* d_base,
                                    const uint64_t* d_total, uint64_t seed, uint32_t frame, int64_t n_total,
                                    int32_t* first)
{
    if (n <= 0) return hipSuccess;
    offspring_offsets_kernel<<<blocks_for(n), kBlock, 0, stream>>>(cdf, n, d_base, d_total, (uint32_t)seed,
                                                                   (uint32_t)(seed >> 32), frame, (uint64_t)n_total,
                                                                   first);
    return hipGetLastError();
}

hipError_t launch_ancestors(hipStream_t stream, const int32_t* first_all, int64_t n_total, int64_t slot0, int nslots,
                            int32_t* anc)
{
    if (nslots <= 0) return hipSuccess;
    ancestors_kernel<<<blocks_for(nslots), kBlock, 0, stream>>>(first_all, n_total, slot0, nslots, anc);
    return hipGetLastError();
}

hipError_t launch_ancestors_sharded(hipStream_t stream, const int32_t* first_all, int64_t n_total, int n, int rank,
                                    int world, int32_t* src)
{
    if (n <= 0) return hipSuccess;
    ancestors_sharded_kernel<<<blocks_for(n), kBlock, 0, stream>>>(first_all, n_total, n, rank, world, src);
    return hipGetLastError();
}

hipError_t launch_migrate_pack(hipStream_t stream, const int32_t* first_all, int64_t n_total, int n, int rank,
                               const MigratePlan& plan, const float* pose, int64_t pose_ld, const float* map,
                               int64_t plane_stride, int ld, int nlandmarks, float* out)
{
    const int total = plan.off[plan.world];
    if (total <= 0) return hipSuccess;
    const int rows = 3 + 5 * nlandmarks;
    migrate_pack_kernel<<<dim3(blocks_for(total), rows < 512 ? rows : 512), kBlock, 0, stream>>>(
        first_all, n_total, n, rank, plan, pose, pose_ld, map, plane_stride, ld, nlandmarks, out);
    return hipGetLastError();
}

hipError_t launch_migrate_unpack(hipStream_t stream, const float* in, const MigratePlan& plan, int n, float* pose,
                                 int64_t pose_ld, float* map, int64_t plane_stride, int ld, int nlandmarks)
{
    const int total = plan.off[plan.world];
    if (total <= 0) return hipSuccess;
    const int rows = 3 + 5 * nlandmarks;
    migrate_unpack_kernel<<<dim3(blocks_for(total), rows < 512 ? rows : 512), kBlock, 0, stream>>>(
        in, plan, n, pose, pose_ld, map, plane_stride, ld, nlandmarks);
    return hipGetLastError();
}

hipError_t launch_argmax(hipStream_t stream, const float* v, int n, int32_t* idx_out, float* val_out)
{
    if (n <= 0) return hipSuccess;
    argmax_kernel<<<1, 1024, 0, stream>>>(v, n, idx_out, val_out);
    return hipGetLastError();
}

hipError_t launch_gather_f32(hipStream_t stream, const float* src, const int32_t* idx, int n, float* dst)
{
    if (n <= 0) return hipSuccess;
    gather_f32_kernel<<<blocks_for(n), kBlock, 0, stream>>>(src, idx, n, dst);
    return hipGetLastError();
}

hipError_t launch_gather_map(hipStream_t stream, const float* in, float* out, int64_t in_stride, int64_t out_stride,
                             int ld_in, int ld_out, int nlandmarks, const int32_t* idx, int n)
{
    if (n <= 0 || nlandmarks <= 0) return hipSuccess;
    const int gy = nlandmarks < 1024 ? nlandmarks : 1024;
    gather_map_kernel<<<dim3(blocks_for(n), gy), kBlock, 0, stream>>>(in, out, in_stride, out_stride, ld_in, ld_out,
                                                                      nlandmarks, idx, n);
    return hipGetLastError();
}

}  // namespace slam
