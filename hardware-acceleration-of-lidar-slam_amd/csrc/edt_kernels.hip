// edt_kernels.hip — capped exact Euclidean distance transform.
//
// Semantics restated from the reference's three formulations
//   gather   Subsystem_1/main.c:223-269
//   scatter  Subsystem_1/main_accelerated.c:215-283  (and Submodule_2/Accelereated_Euclidean_Distance_Transform.c)
// which all produce, over the used rows x cols rectangle only,
//   out = 0                          on occupied cells
//   out = sqrtf((float)min_d2)       if (float)min_d2 < cap*cap   (min over occupied cells of dx^2+dy^2)
//   out = cap                        otherwise
// (a candidate is only ever accepted while d2 < cap^2, so occupied cells farther than
// ceil(cap) in x or y can never matter.)
//
// Mapping to CDNA4: the reference's O(cells x occupied) loops become a separable two-step search in
// LDS.  A workgroup owns a 64 x 16 output tile and needs the (64+2R) x (16+2R) occupancy halo around it.
// Stage: a wavefront loads one halo row per step, 64 + 2R cells as two coalesced wave loads, and turns each
// into a BIT MASK with a ballot — all loads of a wavefront are issued before the first ballot.  Step 1: the
// nearest occupied cell along x within R is two count-zero instructions on the 2R+1-bit window of that
// mask.  Step 2 combines those along y (2R+1 LDS byte reads) with integer d^2 — exact, no float until the
// final correctly-rounded sqrt.  HBM traffic is the algorithmic 4 B in + 4 B out per cell; the halo
// re-reads are L2 hits.  (The first version staged the halo cell by cell as bytes — 12 dependent global
// loads per thread with a division each — and searched it with byte reads: 20 us for a 200 x 200 grid,
// 25 us at 1024^2; it is kept below for R = 32, whose window does not fit 64 bits.)

#include "kernels.h"

namespace slam {

namespace {

constexpr int kTileW = 64;
constexpr int kTileH = 16;
constexpr int kEdtBlock = 256;

// ITER: halo rows per wavefront, ceil((kTileH + 2 R) / 4) rounded to one of two instantiations
template <int ITER>
__global__ __launch_bounds__(kEdtBlock) void edt_bits_kernel(const int32_t* __restrict__ occ, int ld, int rows, int cols,
                                                             float cap, float cap2, int rad, float* __restrict__ out)
{
    extern __shared__ unsigned long long s_bits[];   // [halo_h][2]: bit b of word 0 = cell c0 - rad + b, word 1 = the next 64
    const int halo_h = kTileH + 2 * rad;
    unsigned char* s_gx = reinterpret_cast<unsigned char*>(s_bits + 2 * halo_h);   // [halo_h][kTileW]
    const int c0 = blockIdx.x * kTileW, r0 = blockIdx.y * kTileH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // stage: everything outside the used rectangle counts as free (the reference never looks there).  Loads are
    // unconditional on a clamped address, so all 2 * ITER of them are in flight before the first ballot.
    int va[ITER], vb[ITER];
    const int ca = c0 - rad + lane, cb = ca + 64;
    const bool col_a = ca >= 0 && ca < cols, col_b = lane < 2 * rad && cb >= 0 && cb < cols;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int hr = wave + 4 * it;
        const int r = r0 - rad + hr;
        const bool row_ok = hr < halo_h && r >= 0 && r < rows;
        const size_t base = row_ok ? (size_t)r * ld : 0;
        va[it] = occ[row_ok && col_a ? base + ca : 0];
        vb[it] = occ[row_ok && col_b ? base + cb : 0];
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int hr = wave + 4 * it;
        const int r = r0 - rad + hr;
        const bool row_ok = hr < halo_h && r >= 0 && r < rows;
        const unsigned long long ba = __ballot(row_ok && col_a && va[it] != 0);
        const unsigned long long bb = __ballot(row_ok && col_b && vb[it] != 0);
        if (lane == 0 && hr < halo_h) {
            s_bits[2 * hr] = ba;
            s_bits[2 * hr + 1] = bb;
        }
    }
    __syncthreads();

    // step 1: nearest occupied cell along x for every halo row and tile column: |dx|, or rad + 1 for none.
    // Window = bits tc .. tc + 2 rad of the row mask; its centre (the cell itself) is bit rad.
    const unsigned long long wmask = (1ull << (2 * rad + 1)) - 1ull, lmask = (1ull << rad) - 1ull;
    for (int k = threadIdx.x; k < halo_h * kTileW; k += kEdtBlock) {
        const int hr = k / kTileW, tc = k - hr * kTileW;
        const unsigned long long lo = s_bits[2 * hr], hi = s_bits[2 * hr + 1];
        const unsigned long long w = ((lo >> tc) | (tc ? hi << (64 - tc) : 0ull)) & wmask;
        const unsigned long long right = w >> rad, left = w & lmask;
        const int dr = right ? __builtin_ctzll(right) : rad + 1;
        const int dl = left ? rad - (63 - __builtin_clzll(left)) : rad + 1;
        s_gx[k] = (unsigned char)(dr < dl ? dr : dl);
    }
    __syncthreads();

    // step 2: combine along y with integer squared distances
    const int tc = threadIdx.x & (kTileW - 1);
    const int c = c0 + tc;
    for (int tr = threadIdx.x / kTileW; tr < kTileH; tr += kEdtBlock / kTileW) {
        const int r = r0 + tr;
        if (r >= rows || c >= cols) continue;
        const unsigned char* col = s_gx + (tr + rad) * kTileW + tc;
        int best = 0x7fffffff;
        for (int dy = -rad; dy <= rad; ++dy) {
            const int gx = col[dy * kTileW];
            const int d2 = gx <= rad ? gx * gx + dy * dy : 0x7fffffff;
            best = d2 < best ? d2 : best;
        }
        float v = cap;
        if (best != 0x7fffffff) {
            const float fd2 = (float)best;
            if (fd2 < cap2) v = sqrtf(fd2);   // correctly rounded (NOT __fsqrt_rn: that is the 1-ulp native sqrt); 0 when occupied
        }
        out[(size_t)r * ld + c] = v;
    }
}

// first version, byte halo; used for rad = 32 only (see the header)
__global__ __launch_bounds__(kEdtBlock) void edt_kernel(const int32_t* __restrict__ occ, int ld, int rows, int cols,
                                                        float cap, float cap2, int rad, float* __restrict__ out)
{
    extern __shared__ unsigned char s_mem[];
    const int halo_w = kTileW + 2 * rad;
    const int halo_h = kTileH + 2 * rad;
    unsigned char* s_occ = s_mem;                    // [halo_h][halo_w]
    unsigned char* s_gx = s_mem + halo_h * halo_w;   // [halo_h][kTileW]: |dx| of nearest occupied, rad+1 = none

    const int c0 = blockIdx.x * kTileW, r0 = blockIdx.y * kTileH;

    // stage occupancy; anything outside the used rectangle counts as free (the reference never looks there)
    for (int k = threadIdx.x; k < halo_h * halo_w; k += kEdtBlock) {
        const int hr = k / halo_w, hc = k - hr * halo_w;
        const int r = r0 - rad + hr, c = c0 - rad + hc;
        unsigned char v = 0;
        if (r >= 0 && r < rows && c >= 0 && c < cols) v = occ[(size_t)r * ld + c] != 0;
        s_occ[k] = v;
    }
    __syncthreads();

    // step 1: nearest occupied cell along x, for every halo row and every tile column
    for (int k = threadIdx.x; k < halo_h * kTileW; k += kEdtBlock) {
        const int hr = k / kTileW, tc = k - hr * kTileW;
        const unsigned char* row = s_occ + hr * halo_w + tc + rad;
        int best = rad + 1;
        if (row[0]) {
            best = 0;
        } else {
            for (int d = 1; d <= rad; ++d)
                if (row[-d] | row[d]) {
                    best = d;
                    break;
                }
        }
        s_gx[k] = (unsigned char)best;
    }
    __syncthreads();

    // step 2: combine along y with integer squared distances
    const int tc = threadIdx.x & (kTileW - 1);
    const int c = c0 + tc;
    for (int tr = threadIdx.x / kTileW; tr < kTileH; tr += kEdtBlock / kTileW) {
        const int r = r0 + tr;
        if (r >= rows || c >= cols) continue;
        const unsigned char* col = s_gx + (tr + rad) * kTileW + tc;
        int best = 0x7fffffff;
        for (int dy = -rad; dy <= rad; ++dy) {
            const int gx = col[dy * kTileW];
            if (gx <= rad) {
                const int d2 = gx * gx + dy * dy;
                best = d2 < best ? d2 : best;
            }
        }
        float v = cap;
        if (best != 0x7fffffff) {
            const float fd2 = (float)best;
            if (fd2 < cap2) v = sqrtf(fd2);   // correctly rounded (NOT __fsqrt_rn: that is the 1-ulp native sqrt); 0 when occupied
        }
        out[(size_t)r * ld + c] = v;
    }
}

}  // namespace

hipError_t launch_edt(hipStream_t stream, const int32_t* occ, int ld, int rows, int cols, float cap, float* out,
                      const EventPair* ev)
{
    if (rows <= 0 || cols <= 0) return hipSuccess;
    int rad = (int)ceilf(cap);
    if (rad < 0) rad = 0;
    if (rad > EDT_MAX_RADIUS) return hipErrorInvalidValue;
    const dim3 grid((cols + kTileW - 1) / kTileW, (rows + kTileH - 1) / kTileH);
    const int halo_h = kTileH + 2 * rad;
    if (ev) (void)hipEventRecord(ev->start, stream);
    if (2 * rad + 1 <= 63) {   // the x window fits one 64-bit mask
        const size_t lds = sizeof(unsigned long long) * 2 * (size_t)halo_h + (size_t)halo_h * kTileW;
        if (halo_h <= 4 * 9)
            edt_bits_kernel<9><<<grid, kEdtBlock, lds, stream>>>(occ, ld, rows, cols, cap, cap * cap, rad, out);
        else
            edt_bits_kernel<20><<<grid, kEdtBlock, lds, stream>>>(occ, ld, rows, cols, cap, cap * cap, rad, out);
    } else {
        const size_t lds = (size_t)halo_h * (kTileW + 2 * rad) + (size_t)halo_h * kTileW;
        edt_kernel<<<grid, kEdtBlock, lds, stream>>>(occ, ld, rows, cols, cap, cap * cap, rad, out);
    }
    if (ev) (void)hipEventRecord(ev->stop, stream);
    return hipGetLastError();
}

}  // namespace slam
