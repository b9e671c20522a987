// mapper_kernels.hip — the reference's per-frame front end on the device:
// scan clean-up, world transform, local-map crop, occupancy rasterisation, map append.
//
// Semantics restated from Subsystem_1/main.c (same float operations in the same order, -ffp-contract=off):
//   clean      :71-95    keep beam k unless r < range_min | r > usable; x = r*cosf(a_k), y = r*sinf(a_k) with the
//                        cos/sin table computed ONCE on the host by libm (the reference calls cosf per beam per
//                        frame on the same angles); order-preserving compaction
//   transform  :97-118   tx = (ct*x + st*y) + px ; ty = (-st*x + ct*y) + py, ct/st from the host libm
//   crop       :155-198  bbox of the transformed scan +- border, map points strictly inside, order kept
//   rasterise  :271-354  bbox of the local map, pad 3 pixels, S = (int)roundf(extent/pix)+1, 1-based hit ->
//                        linear index -> (row, col) = 1
//   append     :941-953  j < nhits with hits[j] > 1.5 -> map gets (tx[j], ty[j]), order kept (Q2 indexing)
// All of these are a few thousand elements per frame: one workgroup each, wavefront ballot/prefix for the
// order-preserving compactions; they exist to keep the map, the scan and the grids resident in HBM, not
// for throughput.

#include "det_math.h"
#include "kernels.h"

namespace slam {

namespace {

constexpr int kMapBlock = 1024;

// order-preserving compaction step for one workgroup: returns this thread's output slot (or -1) for the
// current batch and advances `base` by the batch's survivor count
__device__ __forceinline__ int block_compact_slot(bool keep, int& base, int* s_wave /*[kMapBlock/64]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long mask = __ballot(keep);
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(mask);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < kMapBlock / 64; ++w) {
        const int c = s_wave[w];
        before += w < wave ? c : 0;
        total += c;
    }
    __syncthreads();
    const int slot = keep ? base + before + rank : -1;
    base += total;
    return slot;
}

__global__ __launch_bounds__(kMapBlock) void clean_scan_kernel(const float* __restrict__ range,
                                                               const float* __restrict__ cos_tab,
                                                               const float* __restrict__ sin_tab, int nbeams,
                                                               float range_min, float usable, float* __restrict__ bx,
                                                               float* __restrict__ by, int32_t* __restrict__ nscan)
{
    __shared__ int s_wave[kMapBlock / 64];
    int base = 0;
    for (int k0 = 0; k0 < nbeams; k0 += kMapBlock) {
        const int k = k0 + threadIdx.x;
        float r = 0.0f;
        bool keep = false;
        if (k < nbeams) {
            r = range[k];
            keep = !((r < range_min) | (r > usable));   // NaN passes both tests and is kept, as in the reference
        }
        const int slot = block_compact_slot(keep, base, s_wave);
        if (slot >= 0) {
            bx[slot] = r * cos_tab[k];
            by[slot] = r * sin_tab[k];
        }
    }
    if (threadIdx.x == 0) *nscan = base;
}

__global__ __launch_bounds__(kMapBlock) void transform_kernel(const float* __restrict__ bx, const float* __restrict__ by,
                                                              const int32_t* __restrict__ nscan, float px, float py,
                                                              float ct, float st, float* __restrict__ tx,
                                                              float* __restrict__ ty)
{
    const int n = *nscan;
    for (int i = threadIdx.x; i < n; i += kMapBlock) {
        const float x = bx[i], y = by[i];
        tx[i] = (ct * x + st * y) + px;
        ty[i] = (-st * x + ct * y) + py;
    }
}

__device__ __forceinline__ void block_minmax(float& lo_x, float& hi_x, float& lo_y, float& hi_y, float* s_red /*[4*16]*/)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo_x = fminf(lo_x, __shfl_xor(lo_x, o));
        hi_x = fmaxf(hi_x, __shfl_xor(hi_x, o));
        lo_y = fminf(lo_y, __shfl_xor(lo_y, o));
        hi_y = fmaxf(hi_y, __shfl_xor(hi_y, o));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_red[wave] = lo_x; s_red[16 + wave] = hi_x; s_red[32 + wave] = lo_y; s_red[48 + wave] = hi_y;
    }
    __syncthreads();
    for (int w = 0; w < kMapBlock / 64; ++w) {
        lo_x = fminf(lo_x, s_red[w]); hi_x = fmaxf(hi_x, s_red[16 + w]);
        lo_y = fminf(lo_y, s_red[32 + w]); hi_y = fmaxf(hi_y, s_red[48 + w]);
    }
    __syncthreads();
}

// local map = map points strictly inside bbox(scan world points) +- border
__global__ __launch_bounds__(kMapBlock) void crop_kernel(const float* __restrict__ tx, const float* __restrict__ ty,
                                                         const int32_t* __restrict__ nscan, float border,
                                                         const float* __restrict__ mx, const float* __restrict__ my,
                                                         const int32_t* __restrict__ msize, int local_cap,
                                                         float* __restrict__ lx, float* __restrict__ ly,
                                                         int32_t* __restrict__ lsize)
{
    __shared__ float s_red[64];
    __shared__ int s_wave[kMapBlock / 64];
    const int n = *nscan, m = *msize;
    float lo_x = INFINITY, hi_x = -INFINITY, lo_y = INFINITY, hi_y = -INFINITY;
    for (int i = threadIdx.x; i < n; i += kMapBlock) {
        lo_x = fminf(lo_x, tx[i]); hi_x = fmaxf(hi_x, tx[i]);
        lo_y = fminf(lo_y, ty[i]); hi_y = fmaxf(hi_y, ty[i]);
    }
    block_minmax(lo_x, hi_x, lo_y, hi_y, s_red);
    lo_x = lo_x - border; lo_y = lo_y - border; hi_x = hi_x + border; hi_y = hi_y + border;
    int base = 0;
    for (int k0 = 0; k0 < m; k0 += kMapBlock) {
        const int k = k0 + threadIdx.x;
        float qx = 0.0f, qy = 0.0f;
        bool keep = false;
        if (k < m) {
            qx = mx[k]; qy = my[k];
            keep = (qx > lo_x) && (qx < hi_x) && (qy > lo_y) && (qy < hi_y);
        }
        const int slot = block_compact_slot(keep, base, s_wave);
        if (slot >= 0 && slot < local_cap) { lx[slot] = qx; ly[slot] = qy; }
    }
    if (threadIdx.x == 0) *lsize = base < local_cap ? base : local_cap;
}

// one resolution: meta (rows, cols, ld, pixel, min_x, min_y) and the occupancy grid
__global__ __launch_bounds__(kMapBlock) void rasterise_kernel(const float* __restrict__ lx, const float* __restrict__ ly,
                                                              const int32_t* __restrict__ lsize, float pixel, int ld,
                                                              int32_t* __restrict__ grid,
                                                              slam_grid_meta* __restrict__ meta)
{
    __shared__ float s_red[64];
    const int n = *lsize;
    // main.c:272-290 seeds the box from element 0 and scans every element
    float lo_x = INFINITY, hi_x = -INFINITY, lo_y = INFINITY, hi_y = -INFINITY;
    for (int i = threadIdx.x; i < n; i += kMapBlock) {
        lo_x = fminf(lo_x, lx[i]); hi_x = fmaxf(hi_x, lx[i]);
        lo_y = fminf(lo_y, ly[i]); hi_y = fmaxf(hi_y, ly[i]);
    }
    block_minmax(lo_x, hi_x, lo_y, hi_y, s_red);
    lo_x -= (3 * pixel); hi_x += (3 * pixel);
    lo_y -= (3 * pixel); hi_y += (3 * pixel);
    const int nx = (int)round_half_away((hi_x - lo_x) / pixel) + 1;
    const int ny = (int)round_half_away((hi_y - lo_y) / pixel) + 1;
    if (threadIdx.x == 0) {
        meta->rows = ny; meta->cols = nx; meta->ld = ld; meta->pixel = pixel; meta->min_x = lo_x; meta->min_y = lo_y;
    }
    for (int k = threadIdx.x; k < ld * ld; k += kMapBlock) grid[k] = 0;   // main.c:319-320: the whole storage
    __syncthreads();
    if (n <= 0 || nx < 1 || ny < 1 || nx > ld || ny > ld) return;         // host checks the meta and reports
    for (int i = threadIdx.x; i < n; i += kMapBlock) {
        const int hx = (int)round_half_away((lx[i] - lo_x) / pixel) + 1;
        const int hy = (int)round_half_away((ly[i] - lo_y) / pixel) + 1;
        const int lin = ((hy - 1) * nx + hx) - 1;
        const int r = lin / nx, c = lin % nx;
        if (r >= 0 && r < ld && c >= 0 && c < ld) grid[r * ld + c] = 1;
    }
}

// map append (main.c:941-953): order-preserving, capacity-guarded
__global__ __launch_bounds__(kMapBlock) void map_append_kernel(const float* __restrict__ hits, int nhits,
                                                               const float* __restrict__ tx, const float* __restrict__ ty,
                                                               float* __restrict__ mx, float* __restrict__ my,
                                                               int32_t* __restrict__ msize, int map_cap, float threshold)
{
    __shared__ int s_wave[kMapBlock / 64];
    const int m0 = *msize;
    int base = 0;
    for (int j0 = 0; j0 < nhits; j0 += kMapBlock) {
        const int j = j0 + threadIdx.x;
        const bool keep = j < nhits && hits[j] > threshold;   // main.c:943 (1.5)
        const int slot = block_compact_slot(keep, base, s_wave);
        if (slot >= 0 && m0 + slot < map_cap) { mx[m0 + slot] = tx[j]; my[m0 + slot] = ty[j]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) *msize = m0 + base < map_cap ? m0 + base : map_cap;
}

}  // namespace

hipError_t launch_clean_scan(hipStream_t s, const float* range, const float* cos_tab, const float* sin_tab, int nbeams,
                             float range_min, float usable, float* bx, float* by, int32_t* nscan)
{
    clean_scan_kernel<<<1, kMapBlock, 0, s>>>(range, cos_tab, sin_tab, nbeams, range_min, usable, bx, by, nscan);
    return hipGetLastError();
}
hipError_t launch_transform(hipStream_t s, const float* bx, const float* by, const int32_t* nscan, float px, float py,
                            float ct, float st, float* tx, float* ty)
{
    transform_kernel<<<1, kMapBlock, 0, s>>>(bx, by, nscan, px, py, ct, st, tx, ty);
    return hipGetLastError();
}
hipError_t launch_crop(hipStream_t s, const float* tx, const float* ty, const int32_t* nscan, float border,
                       const float* mx, const float* my, const int32_t* msize, int local_cap, float* lx, float* ly,
                       int32_t* lsize)
{
    crop_kernel<<<1, kMapBlock, 0, s>>>(tx, ty, nscan, border, mx, my, msize, local_cap, lx, ly, lsize);
    return hipGetLastError();
}
hipError_t launch_rasterise(hipStream_t s, const float* lx, const float* ly, const int32_t* lsize, float pixel, int ld,
                            int32_t* grid, slam_grid_meta* meta)
{
    rasterise_kernel<<<1, kMapBlock, 0, s>>>(lx, ly, lsize, pixel, ld, grid, meta);
    return hipGetLastError();
}
hipError_t launch_map_append(hipStream_t s, const float* hits, int nhits, const float* tx, const float* ty, float* mx,
                             float* my, int32_t* msize, int map_cap, float threshold)
{
    map_append_kernel<<<1, kMapBlock, 0, s>>>(hits, nhits, tx, ty, mx, my, msize, map_cap, threshold);
    return hipGetLastError();
}

}  // namespace slam
