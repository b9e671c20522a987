// profiles/copy_ceiling.hip — what a PURE COPY with the EKF kernel's access shape reaches on this GPU: one wavefront per
// map row, five planes, 256-byte (dword) or 1-KiB (dwordx4) wave accesses, all loads of a batch before its stores,
// XCD-contiguous workgroup numbering, streaming stores — the addressing of ekf_update_kernel without its arithmetic.
// It bounds what any kernel that reads 20 B and writes 20 B per (particle, landmark) can do: measurement tooling, not
// product code.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o copy_ceiling profiles/copy_ceiling.hip ; ./copy_ceiling [rows] [Lp]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// W = floats per lane per access (1 or 4); NB batches in flight; NT = nt stores
// share > 1: rows i - i % share .. share one SOURCE row (what a resampled population looks like: the offspring of an
// ancestor are neighbours and re-read its row from L2); share == 0: write only (no loads at all)
template <int W, int NB, bool NT>
__global__ __launch_bounds__(256) void copy_rows(const float* __restrict__ in, float* __restrict__ out, int n, int Lp, int xcd_chunk,
                                                 int share = 1)
{
    const unsigned lane = threadIdx.x & 63u;
    const int wave = threadIdx.x >> 6;
    int bid = blockIdx.x;
    if (xcd_chunk > 0) bid = (bid & 7) * xcd_chunk + (bid >> 3);
    const int i = bid * 4 + wave;
    if (i >= n) return;
    const float* rin = in + (size_t)(share > 1 ? i - i % share : i) * 5 * Lp;
    if (share == 0) {   // write only
        float* ro = out + (size_t)i * 5 * Lp;
        for (int lb = 0; lb + 128 <= Lp; lb += 128)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 5; ++p) {
                    float* d = &ro[p * Lp + lb + t * 64 + lane];
                    if (NT) __builtin_nontemporal_store(1.0f, d); else *d = 1.0f;
                }
        return;
    }
    float* rout = out + (size_t)i * 5 * Lp;
    constexpr int B = 64 * W * (W == 1 ? 2 : 1);   // landmarks per batch: 128 (two dwords per lane) or 256 (one x4 per lane)
    for (int lb = 0; lb + B * NB <= Lp; lb += B * NB) {
        if constexpr (W == 1) {
            float m[NB][2][5];
#pragma unroll
            for (int g = 0; g < NB; ++g)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int p = 0; p < 5; ++p) m[g][t][p] = rin[p * Lp + lb + g * 128 + t * 64 + lane];
#pragma unroll
            for (int g = 0; g < NB; ++g)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int p = 0; p < 5; ++p) {
                        float* d = &rout[p * Lp + lb + g * 128 + t * 64 + lane];
                        if (NT) __builtin_nontemporal_store(m[g][t][p] + 1.0f, d); else *d = m[g][t][p] + 1.0f;
                    }
        } else {
            v4f m[NB][5];
#pragma unroll
            for (int g = 0; g < NB; ++g)
#pragma unroll
                for (int p = 0; p < 5; ++p) m[g][p] = *(const v4f*)&rin[p * Lp + lb + g * 256 + 4 * lane];
#pragma unroll
            for (int g = 0; g < NB; ++g)
#pragma unroll
                for (int p = 0; p < 5; ++p) {
                    v4f* d = (v4f*)&rout[p * Lp + lb + g * 256 + 4 * lane];
                    const v4f v = m[g][p] + 1.0f;
                    if (NT) __builtin_nontemporal_store(v, d); else *d = v;
                }
        }
    }
}


// G neighbouring rows per wavefront, the source batch kept in REGISTERS while it is stored G times (what
// ekf_update_group_kernel does with the rows of a repeated ancestor): rows i - i % share .. share a source, share >= G
template <int NB, int G>
__global__ __launch_bounds__(256) void copy_rows_group(const float* __restrict__ in, float* __restrict__ out, int n, int Lp, int xcd_chunk,
                                                       int share)
{
    const unsigned lane = threadIdx.x & 63u;
    const int wave = threadIdx.x >> 6;
    int bid = blockIdx.x;
    if (xcd_chunk > 0) bid = (bid & 7) * xcd_chunk + (bid >> 3);
    const int g0 = (bid * 4 + wave) * G;
    if (g0 >= n) return;
    for (int lb = 0; lb + 128 * NB <= Lp; lb += 128 * NB) {
        float m[NB][2][5];
        int prev = -1;
        for (int k = 0; k < G && g0 + k < n; ++k) {
            const int i = g0 + k, src = share > 1 ? i - i % share : i;
            if (src != prev) {
                const float* rin = in + (size_t)src * 5 * Lp;
#pragma unroll
                for (int g = 0; g < NB; ++g)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int p = 0; p < 5; ++p) m[g][t][p] = rin[p * Lp + lb + g * 128 + t * 64 + lane];
                prev = src;
            }
            float* rout = out + (size_t)i * 5 * Lp;
#pragma unroll
            for (int g = 0; g < NB; ++g)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int p = 0; p < 5; ++p) __builtin_nontemporal_store(m[g][t][p] + (float)k, &rout[p * Lp + lb + g * 128 + t * 64 + lane]);
        }
    }
}

template <class K> float time_it(K launch, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int r = 0; r < 3; ++r) launch(r);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) launch(r);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 65536, Lp = argc > 2 ? atoi(argv[2]) : 512;
    const size_t bytes = (size_t)n * 5 * Lp * 4;
    float *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    const int blocks = (n + 3) / 4, chunk = (blocks + 7) / 8, grid = chunk * 8;
    const double gb = 2.0 * bytes / 1e9;
#define RUN(name, W, NB, NT, inplace)                                                                              \
    {                                                                                                              \
        float ms = time_it([&](int r) {                                                                            \
            float* src = (r & 1) ? b : a; float* dst = inplace ? src : ((r & 1) ? a : b);                          \
            copy_rows<W, NB, NT><<<grid, 256>>>(src, dst, n, Lp, chunk); }, 20);                                    \
        printf("%-34s %8.1f us  %7.0f GB/s (read+write)\n", name, ms * 1e3, gb / (ms * 1e-3));                     \
    }
    printf("rows %d x 5 x %d floats: %.1f MB each way\n", n, Lp, bytes / 1e6);
    RUN("dword  nb1 nt  out-of-place", 1, 1, true, false)
    RUN("dword  nb2 nt  out-of-place", 1, 2, true, false)
    RUN("dword  nb2     out-of-place", 1, 2, false, false)
    RUN("dword  nb2 nt  in-place", 1, 2, true, true)
    RUN("dword  nb2     in-place", 1, 2, false, true)
    RUN("x4     nb1 nt  out-of-place", 4, 1, true, false)
    RUN("x4     nb2 nt  out-of-place", 4, 2, true, false)
    RUN("x4     nb1     out-of-place", 4, 1, false, false)
    RUN("x4     nb1 nt  in-place", 4, 1, true, true)
    RUN("x4     nb2 nt  in-place", 4, 2, true, true)
    RUN("x4     nb1     in-place", 4, 1, false, true)
#define RUNS(name, share, bytes_moved)                                                                             \
    {                                                                                                              \
        float ms = time_it([&](int r) {                                                                            \
            float* src = (r & 1) ? b : a; float* dst = (r & 1) ? a : b;                                            \
            copy_rows<1, 2, true><<<grid, 256>>>(src, dst, n, Lp, chunk, share); }, 20);                            \
        printf("%-34s %8.1f us  %7.0f GB/s (HBM bytes: %s)\n", name, ms * 1e3, (bytes_moved) / 1e9 / (ms * 1e-3), #bytes_moved); \
    }
    RUNS("dword  nb2 nt  write only", 0, (double)bytes)
    RUNS("dword  nb2 nt  source shared by 16", 16, (double)bytes * (1.0 + 1.0 / 16))
    RUNS("dword  nb2 nt  source shared by 64", 64, (double)bytes * (1.0 + 1.0 / 64))
#define RUNG(name, G, share, bytes_moved)                                                                          \
    {                                                                                                              \
        const int gblocks = (n + 4 * G - 1) / (4 * G), gchunk = (gblocks + 7) / 8;                                 \
        float ms = time_it([&](int r) {                                                                            \
            float* src = (r & 1) ? b : a; float* dst = (r & 1) ? a : b;                                            \
            copy_rows_group<2, G><<<gchunk * 8, 256>>>(src, dst, n, Lp, gchunk, share); }, 20);                     \
        printf("%-34s %8.1f us  %7.0f GB/s (HBM bytes: %s)\n", name, ms * 1e3, (bytes_moved) / 1e9 / (ms * 1e-3), #bytes_moved); \
    }
    RUNG("grouped x2, source shared by 16", 2, 16, (double)bytes * (1.0 + 1.0 / 16))
    RUNG("grouped x4, source shared by 16", 4, 16, (double)bytes * (1.0 + 1.0 / 16))
    RUNG("grouped x8, source shared by 16", 8, 16, (double)bytes * (1.0 + 1.0 / 16))
    RUNG("grouped x4, source shared by 4", 4, 4, (double)bytes * (1.0 + 1.0 / 4))
    RUNG("grouped x2, nothing shared", 2, 1, 2.0 * (double)bytes)
    return 0;
}
