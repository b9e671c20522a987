// profiles/page_copy_ceiling.hip — what a PURE COPY of scattered pages reaches on this GPU: every "particle" (a group of
// PL lanes, 64 / PL particles per wavefront) reads T pages named by a random table and writes T pages named by another —
// the memory side of the paged landmark update (csrc/paged_kernels.hip) without tables, stamps or arithmetic.  Pages are
// 5 planes x PL floats (PL = 8, 16, 32: 160, 320, 640 bytes).  It bounds what any kernel that moves whole pages can do:
// measurement tooling, not product code.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o page_copy_ceiling profiles/page_copy_ceiling.hip ; ./page_copy_ceiling [particles] [pool pages of 640 B]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// src[i * T + t], dst[i * T + t]: page numbers; share: neighbouring `share` particles read the same source pages
template <int PL, int T, bool READ, bool WRITE>
__global__ __launch_bounds__(256) void copy_pages(const float* __restrict__ in, float* __restrict__ out, const int* __restrict__ src,
                                                  const int* __restrict__ dst, int n, int share)
{
    constexpr int G = 64 / PL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = (blockIdx.x * 4 + wave) * G + lane / PL, slot = lane % PL;
    if (i >= n) return;
    const int is = share > 1 ? i - i % share : i;
    float v[T][5];
    long o[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const long a = (long)src[(long)is * T + t] * 5 * PL + slot;
        o[t] = (long)dst[(long)i * T + t] * 5 * PL + slot;
#pragma unroll
        for (int p = 0; p < 5; ++p) v[t][p] = READ ? in[a + p * PL] : 1.0f;
    }
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            if (WRITE) out[o[t] + p * PL] = v[t][p] + 1.0f;
            else if (v[t][p] == 123.456f) out[o[t]] = 0.0f;   // keep the loads alive
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

template <int PL, int T> void run(float* pool, int n, long pool_bytes, int scatter)
{
    const long npages = pool_bytes / (5 * PL * 4);
    std::vector<int> perm(npages);
    std::iota(perm.begin(), perm.end(), 0);
    std::mt19937 rng(7);
    if (scatter) std::shuffle(perm.begin(), perm.end(), rng);
    const long need = 2L * n * T;
    if (need > npages) { printf("PL %d T %d: pool too small\n", PL, T); return; }
    int *src, *dst;
    CK(hipMalloc(&src, (long)n * T * 4)); CK(hipMalloc(&dst, (long)n * T * 4));
    CK(hipMemcpy(src, perm.data(), (long)n * T * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dst, perm.data() + (long)n * T, (long)n * T * 4, hipMemcpyHostToDevice));
    constexpr int G = 64 / PL;
    const int grid = (n + 4 * G - 1) / (4 * G);
    const double bytes = (double)n * T * 5 * PL * 4;
    const char* tag = scatter ? "scattered" : "in order ";
    float ms = time_it([&](int) { copy_pages<PL, T, true, true><<<grid, 256>>>(pool, pool, src, dst, n, 1); }, 20);
    printf("%s page %4d B x %d per particle  copy        %7.1f us  %6.0f GB/s (read+write)\n", tag, 5 * PL * 4, T, ms * 1e3, 2 * bytes / 1e9 / (ms * 1e-3));
    ms = time_it([&](int) { copy_pages<PL, T, false, true><<<grid, 256>>>(pool, pool, src, dst, n, 1); }, 20);
    printf("%s page %4d B x %d per particle  write only  %7.1f us  %6.0f GB/s\n", tag, 5 * PL * 4, T, ms * 1e3, bytes / 1e9 / (ms * 1e-3));
    ms = time_it([&](int) { copy_pages<PL, T, true, false><<<grid, 256>>>(pool, pool, src, dst, n, 1); }, 20);
    printf("%s page %4d B x %d per particle  read only   %7.1f us  %6.0f GB/s\n", tag, 5 * PL * 4, T, ms * 1e3, bytes / 1e9 / (ms * 1e-3));
    ms = time_it([&](int) { copy_pages<PL, T, true, true><<<grid, 256>>>(pool, pool, src, dst, n, 16); }, 20);
    printf("%s page %4d B x %d per particle  copy, 16 particles share their source pages %7.1f us  %6.0f GB/s (HBM: write + read/16)\n", tag,
           5 * PL * 4, T, ms * 1e3, bytes * (1 + 1.0 / 16) / 1e9 / (ms * 1e-3));
    CK(hipFree(src)); CK(hipFree(dst));
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 65536;
    const long pool_pages = argc > 2 ? atol(argv[2]) : 2L * 65536 * 16;   // 640-byte pages: the pool of 65536 x 500
    const long pool_bytes = pool_pages * 640;
    float* pool;
    CK(hipMalloc(&pool, pool_bytes));
    CK(hipMemset(pool, 0, pool_bytes));
    printf("particles %d, pool %.2f GB\n", n, pool_bytes / 1e9);
    for (int scatter = 1; scatter >= 0; --scatter) {
        run<32, 5>(pool, n, pool_bytes, scatter);    // what 32 observed of 500 landmarks touch with 32-landmark pages
        run<16, 7>(pool, n, pool_bytes, scatter);    // ... with 16-landmark pages
        run<8, 11>(pool, n, pool_bytes, scatter);    // ... with 8-landmark pages
    }
    return 0;
}
