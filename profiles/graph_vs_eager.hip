// profiles/graph_vs_eager.hip — what would capturing the five-launch frame of a small population in a hipGraph buy?
// Five tiny dependent kernels (the frame of slam_pf_step at <= 4k particles is five launches of a few microseconds each),
// issued (a) eagerly, (b) as one replayed graph, (c) as a replayed graph whose kernel arguments are updated before every
// replay (the frame's arguments change every frame: frame counter, odometry increment, which of the two buffers is
// current), each with and without a host synchronisation per frame (slam_pf_main reads an estimate every frame).
// Build + run:  hipcc -O2 --offload-arch=gfx950 -o /tmp/gve profiles/graph_vs_eager.hip && /tmp/gve
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void stage(const float* __restrict__ in, float* __restrict__ out, int n, float k)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * 0.999f + k;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const int n = 4096, frames = 2000, nk = 5;
    float *a, *b;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto eager_frame = [&](int f) {
        for (int k = 0; k < nk; ++k) stage<<<n / 256, 256, 0, s>>>(k & 1 ? b : a, k & 1 ? a : b, n, (float)f);
    };
    // graph of the same five launches
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    eager_frame(0);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    hipGraphNode_t nodes[16]; CK(hipGraphGetNodes(g, nodes, &nn));
    printf("graph nodes: %zu\n", nn);

    for (int sync = 0; sync < 2; ++sync) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int w = 0; w < 200; ++w) { if (mode == 0) eager_frame(w); else CK(hipGraphLaunch(ge, s)); }
            CK(hipStreamSynchronize(s));
            const double t0 = now_us();
            for (int f = 0; f < frames; ++f) {
                if (mode == 0) eager_frame(f);
                else {
                    if (mode == 2)
                        for (size_t k = 0; k < nn; ++k) {   // every node's arguments change from frame to frame
                            const float* in = k & 1 ? b : a; float* out = k & 1 ? a : b; int nv = n; float kv = (float)f;
                            void* args[] = { &in, &out, &nv, &kv };
                            hipKernelNodeParams p = {};
                            p.func = (void*)stage; p.gridDim = dim3(n / 256); p.blockDim = dim3(256); p.kernelParams = args;
                            CK(hipGraphExecKernelNodeSetParams(ge, nodes[k], &p));
                        }
                    CK(hipGraphLaunch(ge, s));
                }
                if (sync) CK(hipStreamSynchronize(s));
            }
            CK(hipStreamSynchronize(s));
            const double us = (now_us() - t0) / frames;
            printf("%-28s %-22s %7.2f us per frame of %d launches\n",
                   mode == 0 ? "eager" : mode == 1 ? "graph replay" : "graph replay + new arguments",
                   sync ? "host sync every frame" : "no sync (queue ahead)", us, nk);
        }
    }
    return 0;
}
