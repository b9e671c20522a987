// The free list of a paged session (paged_kernels.hip) as a body that a launch of another kernel can carry in workgroups of
// its own: the list depends on the page list of the frame (which asks for it) and on the stamps of the last update only, the
// motion + score launch of the frame depends on neither — so on one GPU the list is made in the shadow of the scorer
// (kernels.h: FreeListRider) instead of costing a launch of its own (2.6 us when there is nothing to do, 22 us when there is).
// No counterpart in the reference (it has no particles or landmarks).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slam {

// PoolState (device, kPoolStateWords int32): bookkeeping of the free list between frames, all of it on the device
enum { kPoolFree = 0,    // entries in the free list
       kPoolUsed = 1,    // ... of which handed out already
       kPoolRenew = 2,   // this frame: the list is made anew before the update takes from it
       kPoolBase = 3,    // this frame: first entry the update takes
       kPoolTicket = 4,  // (unused since round 4)
       kPoolShort = 5,   // set (and never cleared) when a list made anew was shorter than what had been reserved from it
       kPoolAcc = 6 };   // words 6-7, one 64-bit word: free_list_kernel's running {entries (high), workgroups done (low)}; left at zero


// ---- free list = pages without the latest stamp (every update stamps every page its new tables name, so at any time the
// pages in use are exactly those with the stamp of the last update).  Run every frame behind page_list_kernel, it does
// something only when that kernel asked for a new list.  A workgroup counts the free pages of its tile, claims that many
// slots of the list with ONE atomic add and fills them (the order of the list does not matter: page numbers are internal).
constexpr int kFreeTile = 8192;

// The reservation is checked against the finished list by the last workgroup to arrive: tables never name more than half the
// pool, so a new list always holds what a frame takes — if that invariant were ever broken the update would hand out pages
// that are still in use, so the shortfall is reported (pool_state[kPoolShort] and, when given, a word in mapped host memory
// that slam_pf_step turns into SLAM_ERR_CAPACITY) instead of passing silently.
// A wavefront reads 64 consecutive stamps per step (one coalesced 256-byte access; one thread walking 32 consecutive stamps
// took 0.6 ms for a 20-million-page pool) and a ballot gives the free ones in order.
// one workgroup of 256 threads: tile `block` of `nblocks`
__device__ __forceinline__ void free_list_body(const uint32_t* __restrict__ stamp, int npages, uint32_t live,
                                               int32_t* __restrict__ freelist, int32_t* __restrict__ pool_state,
                                               int32_t* __restrict__ h_short, int block, int nblocks)
{
    __shared__ int s_w[4];
    __shared__ int s_base;
    if (pool_state[kPoolRenew] == 0) return;
    constexpr int kSteps = kFreeTile / 256;   // 64-page steps per wavefront
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w0 = block * kFreeTile + wave * (kFreeTile / 4);   // this wavefront's pages: w0 .. w0 + 2047
    unsigned long long mask[kSteps];
    int c = 0;
#pragma unroll
    for (int k = 0; k < kSteps; ++k) {
        const int p = w0 + 64 * k + lane;
        const bool fr = p < npages && stamp[p < npages ? p : npages - 1] != live;
        mask[k] = __ballot(fr);
        c += __popcll(mask[k]);   // wave-uniform
    }
    if (lane == 0) s_w[wave] = c;
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w = 0; w < 4; ++w) {
        woff += w < wave ? s_w[w] : 0;
        tot += s_w[w];
    }
    // ONE 64-bit atomic per workgroup carries its share of the list (high word) and "one more workgroup done" (low word); the
    // last one to arrive knows the length of the list from what the atomic returned.  Nothing a workgroup wrote is read by
    // another, so no fence is needed (round 3 took a ticket behind a __threadfence(): on this part an agent-scope release
    // writes the L2 back, once per workgroup — the launch took 29 us at 2 million pages, profiles/r04_split_tuning.md section 9)
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(pool_state + kPoolAcc);
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        const unsigned long long old = atomicAdd(acc, ((unsigned long long)(unsigned)tot << 32) | 1ull);
        s_base = (int)(old >> 32);
        s_last = (unsigned)(old & 0xffffffffu) == (unsigned)nblocks - 1u ? 1 : 0;
    }
    __syncthreads();
    int out = s_base + woff;
#pragma unroll
    for (int k = 0; k < kSteps; ++k) {
        const unsigned long long m = mask[k];
        if (m >> lane & 1ull) freelist[out + __popcll(m & ((1ull << lane) - 1ull))] = w0 + 64 * k + lane;
        out += __popcll(m);
    }
    if (threadIdx.x == 0 && s_last) {   // every workgroup's share is in
        const int have = s_base + tot;
        pool_state[kPoolFree] = have;
        atomicExch(acc, 0ull);
        if (have < pool_state[kPoolUsed]) {
            pool_state[kPoolShort] = 1;
            if (h_short) *h_short = 1;
        }
    }
}


}  // namespace slam
