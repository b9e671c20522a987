// paged_kernels.hip — landmark maps as copy-on-write PAGES (the K-observed frame).
//
// One row per particle (pf_kernels.hip) makes every resampling frame rewrite every row in full, because the fused gather
// writes particle j's copy of its ancestor's row into the other buffer: 10 KB per particle at 500 landmarks, whether the
// frame observed 500 landmarks or 5.  Here a particle's map is a PAGE TABLE: entry b names the page that holds landmarks
// 32 b ... 32 b + 31 (5 planes x 32 floats = 640 bytes = five 128-byte lines).  Resampling copies page-table rows (4 bytes
// per 32 landmarks); offspring of one ancestor SHARE its pages.  The update of a frame touches only the pages that hold an
// observed landmark: it reads the ancestor's page, applies the update and writes the result to a FRESH page from the free
// list (a shared page is never written), and names that page in the particle's new table.  Pages nobody names any more are
// found once per frame: the update stamps every page the new tables name; the free list of the next frame is the list of
// pages without this frame's stamp (a compaction: count, offsets, scatter).
//
// Arithmetic, its order and the log-likelihood summation order (landmark l adds to accumulator l mod 128 in order of l,
// then j + (j + 64), then the xor butterfly) are those of ekf_batches in pf_kernels.hip: a paged and a row-per-particle
// session give the same bits (tests/test_gpu_paged.py).
// No counterpart in the reference (it has no particles or landmarks).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "det_math.h"
#include "ekf_math.h"
#include "free_list_body.h"
#include "kernels.h"

namespace slam {

namespace {

constexpr int kPage = kPageLandmarks;          // landmarks per page
constexpr int kPageFloats = 5 * kPage;         // floats per page of a session on joint pages: planes mu_x | mu_y | P_xx | P_xy | P_yy

// offset (in floats) of page `page` from the pool's base (kernels.h: PageGeom)
__device__ __forceinline__ int64_t page_off(const PageGeom& g, int64_t page)
{
    return page * (g.planes * kPage) + (page >= g.half_pages ? g.gap : 0);
}
constexpr int kWaves = 4;                      // wavefronts per workgroup (64 / kPage particles each)
// `want` fresh pages for whoever calls (one thread): they come from the part of the free list nobody has been given yet —
// or, when that is too short, from a list made anew by free_list_kernel (launched behind the caller, it looks at the
// flag); such a list always holds at least half the pool.
__device__ __forceinline__ void pool_reserve(int32_t* __restrict__ pool_state, int64_t want)
{
    const bool renew = (int64_t)pool_state[kPoolUsed] + want > (int64_t)pool_state[kPoolFree];
    pool_state[kPoolRenew] = renew ? 1 : 0;
    if (renew) pool_state[kPoolFree] = 0;   // the list kernel adds its tiles' counts to it
    const int first = renew ? 0 : pool_state[kPoolUsed];
    pool_state[kPoolBase] = first;
    pool_state[kPoolUsed] = first + (int)want;
}

// The pages a frame touches: page b is touched when one of its landmarks has an observation (table form: NaN = none).
// One workgroup, one landmark per thread and step: a page is kPage neighbouring lanes of a wavefront, "touched" a ballot.
// tpage[0 .. T) ascending, tindex[b] = position of page b in tpage or -1, count[0] = T.
// What SLAM_MAP_AUTO decides the layout from: votes[0] counts the samples in a row with at most two sevenths of the landmarks
// observed, votes[1] those in a row with more than three eighths (device memory, kept by the kernels that take a sample); the
// mirror in mapped host memory holds {observed, L, seq, votes[0], votes[1]}.
__device__ __forceinline__ void publish_obs_count(int nobs, int L, int32_t* __restrict__ votes, int32_t* __restrict__ h_obs,
                                                  uint32_t seq)
{
    int vp = votes[0], vr = votes[1];
    if (7 * (int64_t)nobs <= 2 * (int64_t)L) {   // (two sevenths: below the measured change-over at every size tried)
        vp = vp < 1000000 ? vp + 1 : vp;
        vr = 0;
    } else if (8 * (int64_t)nobs > 3 * (int64_t)L) {   // (three eighths: r04_split_tuning.md section 8, the measured change-over is at 0.28-0.33)
        vr = vr < 1000000 ? vr + 1 : vr;
        vp = 0;
    } else {
        vp = vr = 0;
    }
    votes[0] = vp;
    votes[1] = vr;
    h_obs[0] = nobs;
    h_obs[1] = L;
    h_obs[3] = vp;
    h_obs[4] = vr;
    __threadfence_system();
    __hip_atomic_store(reinterpret_cast<uint32_t*>(h_obs) + 2, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// h_obs (optional, mapped host memory; then votes != nullptr too): this launch also takes SLAM_MAP_AUTO's sample.
// Besides tpage / tindex / count: tmask[t] = which landmarks of touched page t are observed (bit s = landmark s of the
// page), tbase[t] = how many observations lie in the touched pages before t (tbase[T] = all of them) — the observation
// list of build_obs_list_kernel is sorted by landmark, so the observations of page t are its entries tbase[t] .. tbase[t+1].
__global__ __launch_bounds__(1024) void page_list_kernel(const float* __restrict__ zx, const float* __restrict__ zy, int L,
                                                         int nb, int32_t* __restrict__ tpage, int32_t* __restrict__ tindex,
                                                         int32_t* __restrict__ tmask, int32_t* __restrict__ tbase,
                                                         int32_t* __restrict__ count, int n, int32_t* __restrict__ pool_state,
                                                         int32_t* __restrict__ h_obs, uint32_t seq, int32_t* __restrict__ votes,
                                                         int32_t* __restrict__ h_touched, ObsListOut ol)
{
    // ol.id != nullptr (L <= kObsListMaxLandmarks): the same pass also makes the compact observation list of
    // build_obs_list_kernel (pf_kernels.hip) — ids ascending, measurements, accumulator rounds, {count, highest round}
    __shared__ unsigned s_bits[kObsListMaxLandmarks / 32];
    __shared__ int s_wave[16], s_wobs[16];
    __shared__ int s_base, s_obase, s_max_round;
    if (threadIdx.x == 0) {
        s_base = 0;
        s_obase = 0;
        s_max_round = 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kPerWave = 64 / kPage;                       // pages per wavefront and step
    constexpr unsigned long long kMask = (1ull << kPage) - 1ull;   // (kPage < 64)
    for (int l0 = 0; l0 < nb * kPage; l0 += 1024) {   // 1024 / kPage pages per step
        const int l = l0 + (int)threadIdx.x;
        bool ob = false;
        if (l < L) {
            const float vx = zx[l], vy = zy[l];
            ob = vx == vx && vy == vy;
        }
        float vx = 0.0f, vy = 0.0f;
        if (ob) {
            vx = zx[l];
            vy = zy[l];
        }
        const unsigned long long m = __ballot(ob);
        if (ol.id) {
            if (lane == 0) s_bits[(l0 >> 5) + 2 * wave] = (unsigned)m;
            if (lane == 32) s_bits[(l0 >> 5) + 2 * wave + 1] = (unsigned)(m >> 32);
        }
        int touched_before = 0, touched_mine = 0, touched_all = 0;   // pages of this wavefront: below mine / mine / all
#pragma unroll
        for (int g = 0; g < kPerWave; ++g) {
            const int tg = (m >> (g * kPage) & kMask) != 0ull ? 1 : 0;
            touched_all += tg;
            if (g < lane / kPage) touched_before += tg;
            if (g == lane / kPage) touched_mine = tg;
        }
        if (lane == 0) {
            s_wave[wave] = touched_all;
            s_wobs[wave] = __popcll(m);
        }
        __syncthreads();
        int off = s_base, ooff = s_obase;
        for (int w = 0; w < wave; ++w) {
            off += s_wave[w];
            ooff += s_wobs[w];
        }
        const int b = l0 / kPage + kPerWave * wave + lane / kPage;
        if (lane % kPage == 0 && b < nb) {
            const int t = off + touched_before;
            tindex[b] = touched_mine ? t : -1;
            if (touched_mine) {
                const int g = lane / kPage;
                tpage[t] = b;
                tmask[t] = (int32_t)(uint32_t)(m >> (g * kPage) & kMask);
                tbase[t] = ooff + __popcll(m & ((1ull << (g * kPage)) - 1ull));   // observations in the pages before this one
            }
        }
        if (ol.id && ob) {
            const int k = ooff + __popcll(m & ((1ull << lane) - 1ull));
            ol.id[k] = l;
            ol.zx[k] = vx;
            ol.zy[k] = vy;
            // round: earlier observed landmarks with the same l mod 128 = the same bit of every fourth word below
            int r = 0;
            for (int b2 = l - 128; b2 >= 0; b2 -= 128) r += (int)((s_bits[b2 >> 5] >> (b2 & 31)) & 1u);
            ol.round[k] = r;
            if (r > 0) atomicMax(&s_max_round, r);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0, otot = 0;
            for (int w = 0; w < 16; ++w) {
                tot += s_wave[w];
                otot += s_wobs[w];
            }
            s_base += tot;
            s_obase += otot;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int T = s_base;
        count[0] = T;
        tbase[T] = s_obase;
        if (ol.id) {
            ol.count[0] = s_obase;
            ol.count[1] = s_max_round;
        }
        if (h_touched) *h_touched = T;   // a hint for the host: how many pages the update's launches should stage at a time
        pool_reserve(pool_state, (int64_t)n * T);   // this frame's n * T fresh pages
        if (h_obs) publish_obs_count(s_obase, L, votes, h_obs, seq);
    }
}

// The same sample for a session that is on rows
__global__ __launch_bounds__(1024) void obs_count_kernel(const float* __restrict__ zx, const float* __restrict__ zy, int L,
                                                         int32_t* __restrict__ h_obs, uint32_t seq, int32_t* __restrict__ votes)
{
    __shared__ int s_nobs;
    if (threadIdx.x == 0) s_nobs = 0;
    __syncthreads();
    int c = 0;
    for (int l = threadIdx.x; l < L; l += 1024) {
        const float vx = zx[l], vy = zy[l];
        c += (vx == vx && vy == vy) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_nobs, c);
    __syncthreads();
    if (threadIdx.x == 0) publish_obs_count(s_nobs, L, votes, h_obs, seq);
}

__global__ void pool_reserve_kernel(int32_t* __restrict__ pool_state, int64_t want) { pool_reserve(pool_state, want); }

// Received rows -> fresh pages behind table rows n .. n + total - 1 (one workgroup per record).  The pages get the stamp
// of the last update: they are in use from now on, whatever a free list made before the next update finds.
__global__ __launch_bounds__(256) void migrate_unpack_paged_kernel(const float* __restrict__ in, int total, int n,
                                                                   float* __restrict__ pose, int64_t pose_ld,
                                                                   float* __restrict__ pool, int32_t* __restrict__ pt, int nb,
                                                                   int nlandmarks, const int32_t* __restrict__ freelist,
                                                                   const int32_t* __restrict__ pool_state,
                                                                   uint32_t* __restrict__ stamp, uint32_t live)
{
    const int p = blockIdx.x;
    if (p >= total) return;
    const float* __restrict__ rec = in + (int64_t)(3 + 5 * nlandmarks) * p;
    if (threadIdx.x < 3) pose[threadIdx.x * pose_ld + n + p] = rec[threadIdx.x];
    const int32_t* __restrict__ mine = freelist + pool_state[kPoolBase] + (int64_t)p * nb;
    for (int b = threadIdx.x; b < nb; b += 256) {
        pt[(int64_t)(n + p) * nb + b] = mine[b];
        stamp[mine[b]] = live;
    }
    for (int pl = 0; pl < 5; ++pl)
        for (int l = threadIdx.x; l < nb * kPage; l += 256)   // the tail of the last page: landmarks that do not exist
            pool[(int64_t)mine[l / kPage] * kPageFloats + pl * kPage + l % kPage] =
                l < nlandmarks ? rec[3 + pl * nlandmarks + l] : (pl == 2 ? -1.0f : 0.0f);
}

// ... on split pages: the means on fresh pages of two planes, the covariances as a class of its own (what
// migrate_unpack_split_kernel of split_kernels.hip does with them)
__global__ __launch_bounds__(256) void migrate_unpack_split_pages_kernel(const float* __restrict__ in, int total, int n,
                                                                         float* __restrict__ pose, int64_t pose_ld,
                                                                         float* __restrict__ pool, PageGeom geom, int32_t* __restrict__ pt,
                                                                         int nb, int nlandmarks, const int32_t* __restrict__ freelist,
                                                                         const int32_t* __restrict__ pool_state,
                                                                         uint32_t* __restrict__ stamp, uint32_t live, float* __restrict__ cov,
                                                                         float* __restrict__ covx, int32_t* __restrict__ cls, int Lp, float q,
                                                                         const int32_t* __restrict__ cls_free, int cls_first,
                                                                         uint32_t* __restrict__ cstamp, uint32_t cstamp_now,
                                                                         int32_t* __restrict__ live_list, int32_t* __restrict__ live_cnt)
{
    const int p = blockIdx.x;
    if (p >= total) return;
    const float* __restrict__ rec = in + (int64_t)(3 + 5 * nlandmarks) * p;
    if (threadIdx.x < 3) pose[threadIdx.x * pose_ld + n + p] = rec[threadIdx.x];
    const int32_t* __restrict__ mine = freelist + pool_state[kPoolBase] + (int64_t)p * nb;
    for (int b = threadIdx.x; b < nb; b += 256) {
        pt[(int64_t)(n + p) * nb + b] = mine[b];
        stamp[mine[b]] = live;
    }
    const int c = cls_free[cls_first + p];
    float* __restrict__ cr = cov + (int64_t)c * 3 * Lp;
    float* __restrict__ xr = covx + (int64_t)c * 2 * Lp;
    for (int l = threadIdx.x; l < nb * kPage; l += 256) {   // (nb * kPage = Lp; the tail of the last page: landmarks that do not exist)
        const bool in_row = l < nlandmarks;
        float* __restrict__ pg = pool + page_off(geom, mine[l / kPage]) + l % kPage;
        pg[0] = in_row ? rec[3 + l] : 0.0f;
        pg[kPage] = in_row ? rec[3 + nlandmarks + l] : 0.0f;
        const float pxx = in_row ? rec[3 + 2 * nlandmarks + l] : 1.0f, pxy = in_row ? rec[3 + 3 * nlandmarks + l] : 0.0f,
                    pyy = in_row ? rec[3 + 4 * nlandmarks + l] : 1.0f;
        cr[l] = pxx;
        cr[Lp + l] = pxy;
        cr[2 * Lp + l] = pyy;
        float idet = 1.0f, hl = 0.0f;
        if (in_row && !(pxx < 0.0f)) ekf_det_terms<float>(pxx, pxy, pyy, q, idet, hl);
        xr[l] = idet;
        xr[Lp + l] = hl;
    }
    if (threadIdx.x == 0) {
        cls[n + p] = c;
        cstamp[c] = cstamp_now;
        live_list[atomicAdd(live_cnt, 1)] = c;
    }
}

// HALF a wavefront = one particle (32 lanes = the 32 landmarks of a page).  (1) its new page-table row: the ancestor's
// entries, fresh pages for the touched ones, every named page stamped; (2) the touched pages one after the other.  A
// particle's work is a handful of dependent round trips (ancestor -> table -> page -> store) and little else, so the
// kernel's time is the number of wavefronts times those round trips: two particles per wavefront halve it (one particle
// per wavefront, two pages per pass: 91 us at 64k x 500 with 32 observed).
template <bool SPLIT>
__global__ __launch_bounds__(kWaves * 64) void ekf_paged_kernel(PagedEkfArgs a)
{
    constexpr int kGroups = 64 / kPage;          // particles per wavefront: kPage lanes = the landmarks of a page
    constexpr int kAccPages = 128 / kPage;       // pages per round of the 128 log-likelihood accumulators
    __shared__ float s_acc[kWaves][kGroups][128];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = lane / kPage, slot = lane % kPage;   // `half`: this lane's particle of the wavefront
    const int first_of_wave = ((int)blockIdx.x * kWaves + wave) * kGroups;
    const int i_raw = first_of_wave + half;
    if (first_of_wave >= a.n) return;
    const bool alive = i_raw < a.n;            // the upper half of the last wavefront may have no particle: it stores nothing
    const int i = alive ? i_raw : a.n - 1;
    const int src = a.anc ? a.anc[i] : i;
    const int T = __builtin_amdgcn_readfirstlane(a.count[0]);
    const int32_t* __restrict__ row_in = a.pt_in + (int64_t)src * a.nb;
    int32_t* __restrict__ row_out = a.pt_out + (int64_t)i * a.nb;
    const int fbase = __builtin_amdgcn_readfirstlane(a.pool_state[kPoolBase]);
    const int32_t* __restrict__ fresh = a.freelist + fbase + (int64_t)i * T;   // this particle's T fresh pages
    if (alive)
        for (int b = slot; b < a.nb; b += kPage) {
            const int t = a.tindex[b];
            const int32_t page = t < 0 ? row_in[b] : fresh[t];
            row_out[b] = page;
            a.stamp[page] = a.stamp_now;   // named by a table of this frame (same value from every writer)
        }
    // SPLIT: a page holds the means only; the covariances are the class's (split_kernels.hip), which follows the particle
    int cls = 0;
    if constexpr (SPLIT) {
        cls = a.cls_in[src];
        if (alive && slot == 0) {
            a.cls_out[i] = cls;
            a.cstamp[cls] = a.cstamp_now;
        }
    }
    const float* __restrict__ crow = SPLIT ? a.cov + (int64_t)cls * 3 * a.plane_stride : nullptr;
    const float* __restrict__ xrow = SPLIT ? a.covx + (int64_t)cls * 2 * a.plane_stride : nullptr;
    float st, ct;
    det_sincosf(a.th[i], st, ct);
    const float s = st, c = ct, px = a.x[i], py = a.y[i], q = a.meas_var;
    float* acc = s_acc[wave][half];
#pragma unroll
    for (int k = 0; k < kAccPages; ++k) acc[slot + kPage * k] = 0.0f;
    for (int c0 = 0; c0 < T; c0 += kPage) {   // kPage touched pages at a time: lane t of the group holds what page c0 + t needs
        const int tl = c0 + slot < T ? c0 + slot : T - 1;
        const int my_b = a.tpage[tl];
        const int my_old = row_in[my_b], my_new = fresh[tl];
        const int tc = T - c0 < kPage ? T - c0 : kPage;
        for (int t = 0; t < tc; ++t) {
            const int from = half * kPage + t;
            const int b = __shfl(my_b, from, 64);
            const int l = b * kPage + slot;
            const float* __restrict__ pin = a.pool + page_off(a.geom, __shfl(my_old, from, 64)) + slot;
            float* __restrict__ pout = a.pool + page_off(a.geom, __shfl(my_new, from, 64)) + slot;
            const bool in = l < a.nlandmarks;
            const float zx = in ? a.obs_zx[l] : __builtin_nanf(""), zy = in ? a.obs_zy[l] : __builtin_nanf("");
            const bool ob = zx == zx && zy == zy;
            if constexpr (SPLIT) {   // means from the page, covariances and their determinant terms from the class's rows
                const float mx = pin[0 * kPage], my = pin[1 * kPage];
                const int lc = l < a.plane_stride ? l : 0;
                const float pxx = crow[lc], pxy = crow[a.plane_stride + lc], pyy = crow[2 * a.plane_stride + lc];
                const EkfShared<float> h = ekf_shared_from<float, false>(pxx, pxy, pyy, q, xrow[lc], xrow[a.plane_stride + lc]);
                const EkfParticle<float> v = ekf_particle<float>(h, mx, my, zx, zy, s, c, px, py);
                const bool first = pxx < 0.0f;
                const float r0 = ob ? (first ? v.wx : v.o0) : mx, r1 = ob ? (first ? v.wy : v.o1) : my;
                const float term = first ? 0.0f : v.ll;
                if (alive) {
                    pout[0 * kPage] = r0;
                    pout[1 * kPage] = r1;
                }
                const int k = (b & (kAccPages - 1)) * kPage + slot;
                if (ob) acc[k] = acc[k] + term;
                continue;
            }
            const float mx = pin[0 * kPage], my = pin[1 * kPage], pxx = pin[2 * kPage], pxy = pin[3 * kPage], pyy = pin[4 * kPage];
            const EkfResult<float> u = ekf_update_one<float>(mx, my, pxx, pxy, pyy, zx, zy, s, c, px, py, q);   // one landmark per lane
            const float o0 = u.o0, o1 = u.o1, o2 = u.o2, o3 = u.o3, o4 = u.o4, ll = u.ll, f0 = u.f0, f1 = u.f1;
            const bool first = pxx < 0.0f;
            float r0 = first ? f0 : o0, r1 = first ? f1 : o1, r2 = first ? q : o2, r3 = first ? 0.0f : o3, r4 = first ? q : o4;
            const float term = first ? 0.0f : ll;
            r0 = ob ? r0 : mx;
            r1 = ob ? r1 : my;
            r2 = ob ? r2 : pxx;
            r3 = ob ? r3 : pxy;
            r4 = ob ? r4 : pyy;
            if (alive) {
                pout[0 * kPage] = r0;
                pout[1 * kPage] = r1;
                pout[2 * kPage] = r2;
                pout[3 * kPage] = r3;
                pout[4 * kPage] = r4;
            }
            // accumulator l mod 128, in order of l: a particle's pages come one per pass, in ascending order
            const int k = (b & (kAccPages - 1)) * kPage + slot;
            if (ob) acc[k] = acc[k] + term;
        }
    }
    // the specification's reduction — t[j] = acc[j] + acc[j + 64], then t[j] += t[j ^ s] for s = 1 .. 32 — on kPage lanes: a
    // lane holds t[slot + kPage m]; the steps s < kPage run across the lanes on each of them, the steps s >= kPage pair
    // them inside the lane (a + b is the same float either way round)
    float u[64 / kPage];
#pragma unroll
    for (int m = 0; m < 64 / kPage; ++m) u[m] = acc[slot + kPage * m] + acc[slot + kPage * m + 64];
#pragma unroll
    for (int sft = 1; sft < kPage; sft <<= 1)
#pragma unroll
        for (int m = 0; m < 64 / kPage; ++m) u[m] = u[m] + __shfl_xor(u[m], sft, 64);
#pragma unroll
    for (int w = 1; w < 64 / kPage; w <<= 1)
#pragma unroll
        for (int m = 0; m < 64 / kPage; m += 2 * w) u[m] = u[m] + u[m + w];
    const float total = u[0];
    if (slot == 0 && alive) {
        a.loglik[i] = total;
        if (a.loglik_user) a.loglik_user[i] = total;
    }
}

// ---- the list form of the paged update, touched pages staged in LDS (same bits): (a) the table row; then, PG touched pages
// at a time:
// (b) the pages are loaded — every load of the chunk in flight together — into an LDS image, (c) the observations of these
// pages (entries tbase[c0] .. tbase[c0 + PG] of the sorted list) are updated IN the image, one lane each: five ds reads,
// the arithmetic, five ds writes, and (d) the image is stored to the fresh pages — whole 128-byte lines, every landmark
// written once.  Against the page-wide form: the arithmetic runs once per observation instead of once per lane of every
// touched page, a particle's pages are not fetched one dependent round trip after the other, and nothing is gathered from or
// scattered to global memory.
template <int PG, bool SPLIT>
__global__ __launch_bounds__(kWaves * 64) void ekf_paged_lds_kernel(PagedEkfArgs a)
{
    constexpr int kGroups = 64 / kPage;
    constexpr int PL = SPLIT ? 2 : 5;                // planes per page
    __shared__ float s_img[kWaves][PG][PL][64];      // [page of the chunk][plane][lane]: lane = particle of the wavefront x slot
    __shared__ float s_acc[kWaves][kGroups][128];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = lane / kPage, slot = lane % kPage;
    const int first_of_wave = ((int)blockIdx.x * kWaves + wave) * kGroups;
    const int i_raw = first_of_wave + half;
    if (first_of_wave >= a.n) return;
    const bool alive = i_raw < a.n;
    const int i = alive ? i_raw : a.n - 1;
    const int src = a.anc ? a.anc[i] : i;
    const int T = __builtin_amdgcn_readfirstlane(a.count[0]);
    const int max_round = __builtin_amdgcn_readfirstlane(a.ol.count[1]);
    const int32_t* __restrict__ row_in = a.pt_in + (int64_t)src * a.nb;
    int32_t* __restrict__ row_out = a.pt_out + (int64_t)i * a.nb;
    const int fbase = __builtin_amdgcn_readfirstlane(a.pool_state[kPoolBase]);
    const int32_t* __restrict__ fresh = a.freelist + fbase + (int64_t)i * T;
    const float* __restrict__ pool_in = a.pool;   // pages named by the ancestors' tables: read only
    float* __restrict__ pool_out = a.pool;        // fresh pages: written only (never one of the above)
    float(*img)[PL][64] = s_img[wave];

    if (alive)
        for (int b = slot; b < a.nb; b += kPage) {
            const int t = a.tindex[b];
            const int32_t page = t < 0 ? row_in[b] : fresh[t];
            row_out[b] = page;
            a.stamp[page] = a.stamp_now;
        }
    int cls = 0;
    if constexpr (SPLIT) {   // the class follows the particle and is still in use
        cls = a.cls_in[src];
        if (alive && slot == 0) {
            a.cls_out[i] = cls;
            a.cstamp[cls] = a.cstamp_now;
        }
    }
    const float* __restrict__ crow = SPLIT ? a.cov + (int64_t)cls * 3 * a.plane_stride : nullptr;
    const float* __restrict__ xrow = SPLIT ? a.covx + (int64_t)cls * 2 * a.plane_stride : nullptr;
    float st, ct;
    det_sincosf(a.th[i], st, ct);
    const float s = st, c = ct, px = a.x[i], py = a.y[i], q = a.meas_var;
    float* acc = s_acc[wave][half];
#pragma unroll
    for (int k = 0; k < 128 / kPage; ++k) acc[slot + kPage * k] = 0.0f;

    for (int c0 = 0; c0 < T; c0 += PG) {
        const int tc = T - c0 < PG ? T - c0 : PG;   // wave-uniform
        // what the pages of this chunk need, looked up by the lanes side by side (lane j of a particle: page c0 + j), then every
        // load of the chunk goes out back to back: nothing here waits for one page before it asks for the next
        const int tl = c0 + slot < T ? c0 + slot : T - 1;
        const int my_old = row_in[a.tpage[tl]], my_new = fresh[tl];
        // (b) pages c0 .. c0 + tc - 1 of the ancestor -> image (chunk slots beyond tc re-read the last page: harmless)
        float v[PG][PL];
        int64_t oout[PG];
#pragma unroll
        for (int j = 0; j < PG; ++j) {
            const int from = half * kPage + (j < kPage ? j : kPage - 1);
            const int64_t oin = page_off(a.geom, __shfl(my_old, from, 64)) + slot;
            oout[j] = page_off(a.geom, __shfl(my_new, from, 64)) + slot;
#pragma unroll
            for (int p = 0; p < PL; ++p) v[j][p] = pool_in[oin + p * kPage];
        }
#pragma unroll
        for (int j = 0; j < PG; ++j)
#pragma unroll
            for (int p = 0; p < PL; ++p) img[j][p][lane] = v[j][p];
        __builtin_amdgcn_wave_barrier();
        // (c) the observations that fall into these pages, one lane each
        const int k_lo = __builtin_amdgcn_readfirstlane(a.tbase[c0]), k_hi = __builtin_amdgcn_readfirstlane(a.tbase[c0 + tc]);
        for (int k0 = k_lo; k0 < k_hi; k0 += kPage) {
            const int k = k0 + slot;
            const bool on = k < k_hi;
            const int kk = on ? k : k_lo;
            const int l = a.ol.id[kk], rnd = a.ol.round[kk];
            const float zx = a.ol.zx[kk], zy = a.ol.zy[kk];
            const int j = a.tindex[l / kPage] - c0, at = half * kPage + l % kPage;
            float term;
            if constexpr (SPLIT) {   // means from the image, covariances and their determinant terms from the class's rows
                const float mx = img[j][0][at], my = img[j][1][at];
                const float pxx = crow[l], pxy = crow[a.plane_stride + l], pyy = crow[2 * a.plane_stride + l];
                const EkfShared<float> h = ekf_shared_from<float, false>(pxx, pxy, pyy, q, xrow[l], xrow[a.plane_stride + l]);
                const EkfParticle<float> w = ekf_particle<float>(h, mx, my, zx, zy, s, c, px, py);
                const bool first = pxx < 0.0f;
                term = first ? 0.0f : w.ll;
                if (on) {
                    img[j][0][at] = first ? w.wx : w.o0;
                    img[j][1][at] = first ? w.wy : w.o1;
                }
            } else {
                const float mx = img[j][0][at], my = img[j][1][at], pxx = img[j][2][at], pxy = img[j][3][at], pyy = img[j][4][at];
                const EkfResult<float> u = ekf_update_one<float>(mx, my, pxx, pxy, pyy, zx, zy, s, c, px, py, q);
                const bool first = pxx < 0.0f;
                term = first ? 0.0f : u.ll;
                if (on) {
                    img[j][0][at] = first ? u.f0 : u.o0;
                    img[j][1][at] = first ? u.f1 : u.o1;
                    img[j][2][at] = first ? q : u.o2;
                    img[j][3][at] = first ? 0.0f : u.o3;
                    img[j][4][at] = first ? q : u.o4;
                }
            }
            // accumulator = landmark mod 128, in order of the landmark: round by round (the observations of one accumulator have
            // distinct rounds; the LDS operations of a wavefront execute in order)
            const int ka = l & 127;
            for (int r = 0; r <= max_round; ++r)
                if (on && rnd == r) acc[ka] = acc[ka] + term;
        }
        __builtin_amdgcn_wave_barrier();
        // (d) image -> fresh pages, whole lines
#pragma unroll
        for (int j = 0; j < PG; ++j)
            if (j < tc && alive) {
#pragma unroll
                for (int p = 0; p < PL; ++p) pool_out[oout[j] + p * kPage] = img[j][p][lane];
            }
        __builtin_amdgcn_wave_barrier();
    }
    float u2[64 / kPage];
#pragma unroll
    for (int m = 0; m < 64 / kPage; ++m) u2[m] = acc[slot + kPage * m] + acc[slot + kPage * m + 64];
#pragma unroll
    for (int sft = 1; sft < kPage; sft <<= 1)
#pragma unroll
        for (int m = 0; m < 64 / kPage; ++m) u2[m] = u2[m] + __shfl_xor(u2[m], sft, 64);
#pragma unroll
    for (int w = 1; w < 64 / kPage; w <<= 1)
#pragma unroll
        for (int m = 0; m < 64 / kPage; m += 2 * w) u2[m] = u2[m] + u2[m + w];
    const float total = u2[0];
    if (slot == 0 && alive) {
        a.loglik[i] = total;
        if (a.loglik_user) a.loglik_user[i] = total;
    }
}

// A frame without observations: the tables follow their particles, nothing else moves.  Every page the new tables name
// gets the new stamp, like in an update: "in use" always means "named by the latest generation of tables".
__global__ __launch_bounds__(256) void page_table_gather_kernel(const int32_t* __restrict__ pt_in, int32_t* __restrict__ pt_out,
                                                                int nb, const int32_t* __restrict__ anc, int n,
                                                                uint32_t* __restrict__ stamp, uint32_t stamp_now)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * nb) return;
    const int i = (int)(idx / nb), b = (int)(idx - (int64_t)i * nb);
    const int32_t page = pt_in[(int64_t)(anc ? anc[i] : i) * nb + b];
    pt_out[idx] = page;
    stamp[page] = stamp_now;
}

// ---- free list = pages without the latest stamp (every update stamps every page its new tables name, so at any time the
// pages in use are exactly those with the stamp of the last update).  Launched every frame behind page_list_kernel, it
// does something only when that kernel asked for a new list (a few times per nb / T frames: a list holds >= n * nb pages
// and a frame takes n * T).  A workgroup counts the free pages of its tile, claims that many slots of the list with ONE
// atomic add and fills them (the order of the list does not matter: page numbers are internal, results do not depend on
// them).
__global__ __launch_bounds__(256) void free_list_kernel(const uint32_t* __restrict__ stamp, int npages, uint32_t live,
                                                        int32_t* __restrict__ freelist, int32_t* __restrict__ pool_state,
                                                        int32_t* __restrict__ h_short)
{
    free_list_body(stamp, npages, live, freelist, pool_state, h_short, (int)blockIdx.x, (int)gridDim.x);
}

// ---- rows <-> pages (set / get of whole maps; not on the frame path)
// rows [n][5][plane_stride] -> pages page_base + j * nb + b (the identity table, shifted)
__global__ __launch_bounds__(256) void pages_from_rows_kernel(const float* __restrict__ rows, int64_t row_stride, int plane_stride,
                                                              int nlandmarks, int nb, int n, float* __restrict__ pool,
                                                              int32_t* __restrict__ pt, int page_base, PageGeom geom)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one thread per (particle, page, slot)
    if (idx >= (int64_t)n * nb * kPage) return;
    const int slot = (int)(idx % kPage);
    const int64_t pb = idx / kPage;
    const int b = (int)(pb % nb), i = (int)(pb / nb);
    const int l = b * kPage + slot;
    const float* r = rows + (int64_t)i * row_stride;
    float* pg = pool + page_off(geom, pb + page_base) + slot;
    for (int p = 0; p < geom.planes; ++p) pg[p * kPage] = l < nlandmarks ? r[(int64_t)p * plane_stride + l] : (p == 2 ? -1.0f : 0.0f);
    if (slot == 0) pt[pb] = (int32_t)pb + page_base;
}

// pages of particle anc[i] (or i) -> row i
__global__ __launch_bounds__(256) void rows_from_pages_kernel(const float* __restrict__ pool, const int32_t* __restrict__ pt,
                                                              int nb, const int32_t* __restrict__ anc, int n,
                                                              float* __restrict__ rows, int64_t row_stride, int plane_stride,
                                                              int nlandmarks, PageGeom geom)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * nb * kPage) return;
    const int slot = (int)(idx % kPage);
    const int64_t pb = idx / kPage;
    const int b = (int)(pb % nb), i = (int)(pb / nb);
    const int l = b * kPage + slot;
    if (l >= nlandmarks) return;
    const int src = anc ? anc[i] : i;
    const float* pg = pool + page_off(geom, pt[(int64_t)src * nb + b]) + slot;
    float* r = rows + (int64_t)i * row_stride;
    for (int p = 0; p < geom.planes; ++p) r[(int64_t)p * plane_stride + l] = pg[p * kPage];
}

// split session on pages -> rows of five planes (one workgroup per output row)
__global__ __launch_bounds__(256) void rows_from_split_pages_kernel(const float* __restrict__ pool, PageGeom geom,
                                                                    const int32_t* __restrict__ pt, int nb, const float* __restrict__ cov,
                                                                    const int32_t* __restrict__ cls, int Lp, const int32_t* __restrict__ idx,
                                                                    int count, float* __restrict__ rows, int64_t row_stride,
                                                                    int plane_stride, int nlandmarks)
{
    const int k = blockIdx.x;
    if (k >= count) return;
    const int src = idx ? idx[k] : k;
    const int32_t* tab = pt + (int64_t)src * nb;
    const float* cr = cov + (int64_t)cls[src] * 3 * Lp;
    float* r = rows + (int64_t)k * row_stride;
    for (int l = threadIdx.x; l < nlandmarks; l += 256) {
        const float* pg = pool + page_off(geom, tab[l / kPage]) + l % kPage;
        r[l] = pg[0];
        r[(int64_t)plane_stride + l] = pg[kPage];
        r[2 * (int64_t)plane_stride + l] = cr[l];
        r[3 * (int64_t)plane_stride + l] = cr[Lp + l];
        r[4 * (int64_t)plane_stride + l] = cr[2 * Lp + l];
    }
}

// every particle starts on ONE shared page of landmarks "not seen yet" (page 0), the rest of the pool is free
__global__ __launch_bounds__(256) void pages_reset_kernel(float* __restrict__ pool, int32_t* __restrict__ pt, int64_t nentries,
                                                          int32_t* __restrict__ freelist, int npages,
                                                          int32_t* __restrict__ pool_state, int planes)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < planes * kPage) pool[idx] = (idx / kPage == 2) ? -1.0f : 0.0f;
    if (idx < nentries) pt[idx] = 0;
    if (idx < npages - 1) freelist[idx] = (int32_t)idx + 1;
    if (idx == 0) {
        pool_state[kPoolFree] = npages - 1;
        pool_state[kPoolUsed] = 0;
        pool_state[kPoolRenew] = 0;
        pool_state[kPoolBase] = 0;
        pool_state[kPoolTicket] = 0;
        pool_state[kPoolShort] = 0;
        pool_state[kPoolAcc] = pool_state[kPoolAcc + 1] = 0;
    }
}

// free list = pages 0 .. count0 - 1, then first1 .. first1 + count1 - 1; nothing handed out
__global__ __launch_bounds__(256) void free_iota_kernel(int32_t* __restrict__ out, int count0, int first1, int count1,
                                                        int32_t* __restrict__ pool_state)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < count0) out[idx] = idx;
    else if (idx < count0 + count1) out[idx] = first1 + (idx - count0);
    if (idx == 0) {
        const int count = count0 + count1;
        pool_state[kPoolFree] = count;
        pool_state[kPoolUsed] = 0;
        pool_state[kPoolRenew] = 0;
        pool_state[kPoolBase] = 0;
        pool_state[kPoolTicket] = 0;
        pool_state[kPoolShort] = 0;
        pool_state[kPoolAcc] = pool_state[kPoolAcc + 1] = 0;
    }
}

// out[k] = anc[sel[k]] (or sel[k]): the source rows of a few chosen particles with the pending gather applied
__global__ __launch_bounds__(256) void compose_index_kernel(const int32_t* __restrict__ sel, const int32_t* __restrict__ anc,
                                                            int count, int32_t* __restrict__ out)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < count) out[k] = anc ? anc[sel[k]] : sel[k];
}

inline int blocks256(int64_t n) { return (int)((n + 255) / 256); }

}  // namespace

hipError_t launch_page_list(hipStream_t stream, const float* zx, const float* zy, int L, int nb, int32_t* tpage, int32_t* tindex,
                            int32_t* tmask, int32_t* tbase, int32_t* count, int n, int32_t* pool_state, int32_t* h_obs, uint32_t seq,
                            int32_t* votes, int32_t* h_touched, const ObsListOut& ol)
{
    ObsListOut o = ol;
    if (L > kObsListMaxLandmarks) o.id = nullptr;
    page_list_kernel<<<1, 1024, 0, stream>>>(zx, zy, L, nb, tpage, tindex, tmask, tbase, count, n, pool_state,
                                             votes ? h_obs : nullptr, seq, votes, h_touched, o);
    return hipGetLastError();
}

hipError_t launch_obs_count(hipStream_t stream, const float* zx, const float* zy, int L, int32_t* h_obs, uint32_t seq, int32_t* votes)
{
    obs_count_kernel<<<1, 1024, 0, stream>>>(zx, zy, L, h_obs, seq, votes);
    return hipGetLastError();
}

hipError_t launch_ekf_paged(hipStream_t stream, const PagedEkfArgs& a, const EventPair* ev, int form, int touched_hint)
{
    if (a.n <= 0) return hipSuccess;
    if (ev) (void)hipEventRecord(ev->start, stream);
    constexpr int per_block = kWaves * (64 / kPage);
    const int grid = (a.n + per_block - 1) / per_block;
    static const int pg_env = getenv("SLAM_PAGED_PG") ? atoi(getenv("SLAM_PAGED_PG")) : 0;
    // pages staged per pass: what the last frame touched (a frame's touched pages then go through in one pass without idle
    // slots); any value gives the same results.  Frames that touch more than six pages per particle run the page-wide form:
    // with that many pages the LDS image costs more wavefronts in flight than the list form saves (measured, 64k x 500 with
    // 128 observed / 64k x 5000 with 512: 120 / 460 us page-wide against 123 / 474 us at best).
    const int pg = pg_env > 0 ? pg_env : (touched_hint > 0 ? touched_hint : 6);
    const bool wide = form == 0 || !a.ol.id || (pg_env <= 0 && touched_hint > 6);
#define SLAM_PAGED(SP_)                                                                              \
    do {                                                                                             \
        if (wide) ekf_paged_kernel<SP_><<<grid, kWaves * 64, 0, stream>>>(a);                        \
        else if (pg <= 1) ekf_paged_lds_kernel<1, SP_><<<grid, kWaves * 64, 0, stream>>>(a);         \
        else if (pg == 2) ekf_paged_lds_kernel<2, SP_><<<grid, kWaves * 64, 0, stream>>>(a);         \
        else if (pg == 3) ekf_paged_lds_kernel<3, SP_><<<grid, kWaves * 64, 0, stream>>>(a);         \
        else if (pg == 4) ekf_paged_lds_kernel<4, SP_><<<grid, kWaves * 64, 0, stream>>>(a);         \
        else if (pg == 5) ekf_paged_lds_kernel<5, SP_><<<grid, kWaves * 64, 0, stream>>>(a);         \
        else if (pg == 6) ekf_paged_lds_kernel<6, SP_><<<grid, kWaves * 64, 0, stream>>>(a);         \
        else ekf_paged_lds_kernel<8, SP_><<<grid, kWaves * 64, 0, stream>>>(a);                      \
    } while (0)
    if (a.geom.planes == 2) SLAM_PAGED(true);
    else SLAM_PAGED(false);
#undef SLAM_PAGED
    if (ev) (void)hipEventRecord(ev->stop, stream);
    return hipGetLastError();
}

hipError_t launch_page_table_gather(hipStream_t stream, const int32_t* pt_in, int32_t* pt_out, int nb, const int32_t* anc, int n,
                                    uint32_t* stamp, uint32_t stamp_now)
{
    if (n <= 0 || nb <= 0) return hipSuccess;
    page_table_gather_kernel<<<blocks256((int64_t)n * nb), 256, 0, stream>>>(pt_in, pt_out, nb, anc, n, stamp, stamp_now);
    return hipGetLastError();
}

int pool_state_words() { return 8; }
int free_list_blocks(int npages) { return (npages + kFreeTile - 1) / kFreeTile; }

hipError_t launch_compose_index(hipStream_t stream, const int32_t* sel, const int32_t* anc, int count, int32_t* out)
{
    if (count <= 0) return hipSuccess;
    compose_index_kernel<<<blocks256(count), 256, 0, stream>>>(sel, anc, count, out);
    return hipGetLastError();
}

hipError_t launch_free_list(hipStream_t stream, const uint32_t* stamp, int npages, uint32_t live, int32_t* freelist,
                            int32_t* pool_state, int32_t* h_short)
{
    free_list_kernel<<<(npages + kFreeTile - 1) / kFreeTile, 256, 0, stream>>>(stamp, npages, live, freelist, pool_state, h_short);
    return hipGetLastError();
}

hipError_t launch_pool_reserve(hipStream_t stream, int32_t* pool_state, int64_t want)
{
    pool_reserve_kernel<<<1, 1, 0, stream>>>(pool_state, want);
    return hipGetLastError();
}

hipError_t launch_migrate_unpack_paged(hipStream_t stream, const float* in, int total, int n, float* pose, int64_t pose_ld,
                                       float* pool, int32_t* pt, int nb, int nlandmarks, const int32_t* freelist,
                                       const int32_t* pool_state, uint32_t* stamp, uint32_t live)
{
    if (total <= 0) return hipSuccess;
    migrate_unpack_paged_kernel<<<total, 256, 0, stream>>>(in, total, n, pose, pose_ld, pool, pt, nb, nlandmarks, freelist,
                                                          pool_state, stamp, live);
    return hipGetLastError();
}

hipError_t launch_migrate_unpack_split_pages(hipStream_t stream, const float* in, int total, int n, float* pose, int64_t pose_ld,
                                             float* pool, const PageGeom& geom, int32_t* pt, int nb, int nlandmarks,
                                             const int32_t* freelist, const int32_t* pool_state, uint32_t* stamp, uint32_t live,
                                             float* cov, float* covx, int32_t* cls, int Lp, float meas_var, const int32_t* cls_free,
                                             int cls_first, uint32_t* cstamp, uint32_t cstamp_now, int32_t* live_list, int32_t* live_cnt)
{
    if (total <= 0) return hipSuccess;
    migrate_unpack_split_pages_kernel<<<total, 256, 0, stream>>>(in, total, n, pose, pose_ld, pool, geom, pt, nb, nlandmarks, freelist,
                                                                 pool_state, stamp, live, cov, covx, cls, Lp, meas_var, cls_free, cls_first,
                                                                 cstamp, cstamp_now, live_list, live_cnt);
    return hipGetLastError();
}

hipError_t launch_pages_from_rows(hipStream_t stream, const float* rows, int64_t row_stride, int plane_stride, int nlandmarks,
                                  int nb, int n, float* pool, int32_t* pt, int32_t* freelist, int npages, int32_t* pool_state,
                                  int page_base, const PageGeom& geom)
{
    pages_from_rows_kernel<<<blocks256((int64_t)n * nb * kPage), 256, 0, stream>>>(rows, row_stride, plane_stride, nlandmarks, nb, n,
                                                                                  pool, pt, page_base, geom);
    const int used = n * nb;   // the tables name pages page_base .. page_base + n * nb - 1: the rest is free
    const int after = page_base + used;
    free_iota_kernel<<<blocks256(npages > used ? npages - used : 1), 256, 0, stream>>>(freelist, page_base, after, npages - after,
                                                                                     pool_state);
    return hipGetLastError();
}

hipError_t launch_rows_from_pages(hipStream_t stream, const float* pool, const int32_t* pt, int nb, const int32_t* anc, int n,
                                  float* rows, int64_t row_stride, int plane_stride, int nlandmarks, const PageGeom& geom)
{
    rows_from_pages_kernel<<<blocks256((int64_t)n * nb * kPage), 256, 0, stream>>>(pool, pt, nb, anc, n, rows, row_stride,
                                                                                  plane_stride, nlandmarks, geom);
    return hipGetLastError();
}

hipError_t launch_rows_from_split_pages(hipStream_t stream, const float* pool, const PageGeom& geom, const int32_t* pt, int nb,
                                        const float* cov, const int32_t* cls, int Lp, const int32_t* idx, int count, float* rows,
                                        int64_t row_stride, int plane_stride, int nlandmarks)
{
    if (count <= 0) return hipSuccess;
    rows_from_split_pages_kernel<<<count, 256, 0, stream>>>(pool, geom, pt, nb, cov, cls, Lp, idx, count, rows, row_stride, plane_stride,
                                                           nlandmarks);
    return hipGetLastError();
}

hipError_t launch_pages_reset(hipStream_t stream, float* pool, int32_t* pt, int64_t nentries, int32_t* freelist, int npages,
                              int32_t* pool_state, const PageGeom& geom)
{
    const int64_t m = nentries > npages ? nentries : npages;
    pages_reset_kernel<<<blocks256(m > kPageFloats ? m : kPageFloats), 256, 0, stream>>>(pool, pt, nentries, freelist, npages,
                                                                                     pool_state, geom.planes);
    return hipGetLastError();
}

}  // namespace slam
