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
// LDS.  A workgroup owns a 64 x 16 output tile, stages the (64+2R) x (16+2R) occupancy halo as
// bytes, step 1 finds for every halo row the nearest occupied cell along x within R (19-21 byte
// reads), step 2 combines those along y (another 2R+1 reads) with integer d^2 — exact, no float
// until the final correctly-rounded sqrt.  HBM traffic is the algorithmic 4 B in + 4 B out per
// cell; at <= 2048^2 the whole job is a few microseconds, i.e. launch-bound.

#include "kernels.h"

namespace slam {

namespace {

constexpr int kTileW = 64;
constexpr int kTileH = 16;
constexpr int kEdtBlock = 256;

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
    const size_t lds = (size_t)(kTileH + 2 * rad) * (kTileW + 2 * rad) + (size_t)(kTileH + 2 * rad) * kTileW;
    if (ev) (void)hipEventRecord(ev->start, stream);
    edt_kernel<<<grid, kEdtBlock, lds, stream>>>(occ, ld, rows, cols, cap, cap * cap, rad, out);
    if (ev) (void)hipEventRecord(ev->stop, stream);
    return hipGetLastError();
}

}  // namespace slam
