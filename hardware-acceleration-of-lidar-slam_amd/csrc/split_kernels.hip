// split_kernels.hip — the SPLIT layout of the landmark maps (no counterpart in the reference, which has
// no landmarks and no covariance anywhere: SURVEY section 0 F1/F2).
//
// The landmark update is specified in the world frame (csrc/ekf_math.h, oracle/slam_oracle_pf.c): S = P + q I, W = P S^-1,
// P' = (I - W) P.  Nothing in P' depends on the particle's pose or on the measurement — only on the prior P, on q and on
// WHETHER the frame observes the landmark, and the observation table is the same for every particle.  So two particles whose
// covariances are equal stay equal for ever, and in a running filter nearly all are: the offspring of an ancestor inherit its
// covariances bit for bit, and a population whose maps were initialised alike never had different ones.  Storing 3 of the 5
// planes of every row per particle means writing the same 12 bytes per landmark over and over.
//
// Here a particle's row holds its MEANS only (two planes), and the covariance planes exist once per COVARIANCE CLASS:
//     mean[particle][2][Lp]     two buffers, the update writes the other one (resample gather fused, as with rows)
//     cls[particle]             the particle's class, handed from ancestor to offspring by the update itself
//     cov[class][3][Lp]         updated IN PLACE, once per class and frame, by cov_update_kernel — after the particles' update,
//                               which reads the prior
//     covx[class][2][Lp]        1 / det (P + q I) and 0.5 log det (P + q I) of the same covariances: the two expensive values
//                               of the update (a reciprocal and a logarithm), worked out by whoever writes cov so that the
//                               particles' update — every wavefront of it — starts from them instead of recomputing them
//     live[..], cnt[3]          the classes still in use, as a list that only ever shrinks: classes die with their last
//                               particle and are never born (set_map / reset start a new epoch)
// The values are those of the row layout, bit for bit (the same ekf_shared / ekf_particle, the same order); the layout is not
// part of the specification.  HBM traffic of a dense frame: 8 B read + 8 B written per (particle, landmark) + 24 B per (class
// in use, landmark), against 20 + 20.
//
// This file: the classes' update, the conversions rows <-> split, the gather of a frame without an update.  The particles'
// update on this layout is ekf_split_body in pf_kernels.hip (it shares the grouped row kernel's machinery).

#include "ekf_math.h"
#include "cov_update_body.h"
#include "kernels.h"

namespace slam {

namespace {

inline int blocks256(int64_t n) { return (int)((n + 255) / 256); }

// ---- the classes' update: workgroup (k, y) takes landmarks [256 y, 256 y + 256) of class live[k] (cov_update_body.h; the frame
// path carries it in the launch of the weights, this launch serves everybody else)
__global__ __launch_bounds__(256) void cov_update_kernel(CovArgs a) { cov_update_body(a, (int)blockIdx.x, (int)blockIdx.y); }

// the determinant terms of classes 0 .. *count - 1 from their covariance planes (after a conversion or a reset)
__global__ __launch_bounds__(256) void cov_terms_kernel(const float* __restrict__ cov, float* __restrict__ covx, int Lp, int nlandmarks,
                                                        float q, const int32_t* __restrict__ count)
{
    const int c = blockIdx.x;
    if (c >= *count) return;
    const int l = blockIdx.y * 256 + threadIdx.x;
    if (l >= Lp) return;
    const float* r = cov + (int64_t)c * 3 * Lp + l;
    float idet = 1.0f, hl = 0.0f;   // (padding columns and landmarks not seen yet: never read)
    if (l < nlandmarks && !(r[0] < 0.0f)) ekf_det_terms<float>(r[0], r[Lp], r[2 * (int64_t)Lp], q, idet, hl);
    float* x = covx + (int64_t)c * 2 * Lp + l;
    x[0] = idet;
    x[Lp] = hl;
}

// ---- rows -> split
// flag[i] = 1 when particle i starts a new class: its covariance planes differ somewhere from its left neighbour's
__global__ __launch_bounds__(256) void cov_heads_kernel(const float* __restrict__ rows, int64_t row_stride, int plane_stride,
                                                        int nlandmarks, int n, uint64_t* __restrict__ flag)
{
    const int i = blockIdx.x;
    if (i >= n) return;
    __shared__ int s_diff;
    if (threadIdx.x == 0) s_diff = i == 0 ? 1 : 0;
    __syncthreads();
    if (i > 0) {
        const uint32_t* a = reinterpret_cast<const uint32_t*>(rows + (int64_t)i * row_stride + 2 * (int64_t)plane_stride);
        const uint32_t* b = reinterpret_cast<const uint32_t*>(rows + (int64_t)(i - 1) * row_stride + 2 * (int64_t)plane_stride);
        bool diff = false;
        for (int p = 0; p < 3; ++p)
            for (int l = threadIdx.x; l < nlandmarks; l += 256) diff = diff || a[(int64_t)p * plane_stride + l] != b[(int64_t)p * plane_stride + l];
        if (diff) s_diff = 1;   // benign race: every writer writes 1
    }
    __syncthreads();
    if (threadIdx.x == 0) flag[i] = (uint64_t)s_diff;
}

// rank[i] = (inclusive prefix sum of flag)[i] - 1 = the class of particle i.  One workgroup per particle: its means, its
// class, and — when it starts a class — the class's covariance row, list entry and stamp.  Padding columns [L, Lp): means 0,
// covariances (1, 0, 1) — "seen, never observed again": harmless operands that nobody reads back.
__global__ __launch_bounds__(256) void split_from_rows_kernel(const float* __restrict__ rows, int64_t row_stride, int plane_stride,
                                                              int nlandmarks, int n, int Lp, const uint64_t* __restrict__ flag,
                                                              const uint64_t* __restrict__ sum, float* __restrict__ mean,
                                                              float* __restrict__ cov, int32_t* __restrict__ cls,
                                                              int32_t* __restrict__ live, int32_t* __restrict__ cnt, int phase,
                                                              uint32_t* __restrict__ cstamp, uint32_t stamp_now,
                                                              int32_t* __restrict__ h_live, uint32_t epoch)
{
    const int i = blockIdx.x;
    if (i >= n) return;
    const int c = (int)sum[i] - 1;
    const bool head = flag[i] != 0;
    const float* r = rows + (int64_t)i * row_stride;
    float* m = mean + (int64_t)i * 2 * Lp;
    for (int l = threadIdx.x; l < Lp; l += 256) {
        const bool in = l < nlandmarks;
        m[l] = in ? r[l] : 0.0f;
        m[Lp + l] = in ? r[(int64_t)plane_stride + l] : 0.0f;
    }
    if (head) {
        float* cr = cov + (int64_t)c * 3 * Lp;
        for (int l = threadIdx.x; l < Lp; l += 256) {
            const bool in = l < nlandmarks;
            cr[l] = in ? r[2 * (int64_t)plane_stride + l] : 1.0f;
            cr[Lp + l] = in ? r[3 * (int64_t)plane_stride + l] : 0.0f;
            cr[2 * Lp + l] = in ? r[4 * (int64_t)plane_stride + l] : 1.0f;
        }
    }
    if (threadIdx.x == 0) {
        cls[i] = c;
        if (head) {
            live[c] = c;
            cstamp[c] = stamp_now;
        }
        if (i == n - 1) {
            cnt[phase] = c + 1;
            cnt[(phase + 1) % 3] = 0;
            cnt[(phase + 2) % 3] = 0;
            if (h_live) publish_live(h_live, c + 1, epoch);
        }
    }
}

// ---- split -> rows: out row k = [means of particle src | covariances of its class], src = idx[k] or k
__global__ __launch_bounds__(256) void rows_from_split_kernel(const float* __restrict__ mean, const float* __restrict__ cov,
                                                              const int32_t* __restrict__ cls, int Lp, const int32_t* __restrict__ idx,
                                                              int count, float* __restrict__ rows, int64_t row_stride,
                                                              int plane_stride, int nlandmarks)
{
    const int k = blockIdx.x;
    if (k >= count) return;
    const int src = idx ? idx[k] : k;
    const float* m = mean + (int64_t)src * 2 * Lp;
    const float* cr = cov + (int64_t)cls[src] * 3 * Lp;
    float* r = rows + (int64_t)k * row_stride;
    for (int l = threadIdx.x; l < nlandmarks; l += 256) {
        r[l] = m[l];
        r[(int64_t)plane_stride + l] = m[Lp + l];
        r[2 * (int64_t)plane_stride + l] = cr[l];
        r[3 * (int64_t)plane_stride + l] = cr[Lp + l];
        r[4 * (int64_t)plane_stride + l] = cr[2 * Lp + l];
    }
}

// ---- a frame without a landmark update: means and classes follow their particles
__global__ __launch_bounds__(256) void split_gather_kernel(const float* __restrict__ mean_in, float* __restrict__ mean_out,
                                                           const int32_t* __restrict__ cls_in, int32_t* __restrict__ cls_out, int Lp,
                                                           const int32_t* __restrict__ anc, int n, uint32_t* __restrict__ cstamp,
                                                           uint32_t stamp_now)
{
    const int i = blockIdx.x;
    if (i >= n) return;
    const int src = anc ? anc[i] : i;
    const float4* s = reinterpret_cast<const float4*>(mean_in + (int64_t)src * 2 * Lp);   // Lp is a multiple of 32 floats
    float4* d = reinterpret_cast<float4*>(mean_out + (int64_t)i * 2 * Lp);
    for (int v = threadIdx.x; v < Lp / 2; v += 256) d[v] = s[v];
    if (threadIdx.x == 0) {
        const int c = cls_in[src];
        cls_out[i] = c;
        cstamp[c] = stamp_now;
    }
}

// ---- class numbers for arrivals (sharded sessions): free = every class whose stamp is older than min_live, in no particular
// order; fs = {fill cursor, ticket}, zero on entry, zeroed again by the last workgroup
__global__ __launch_bounds__(256) void class_free_list_kernel(const uint32_t* __restrict__ cstamp, int nclasses, uint32_t min_live,
                                                              int32_t* __restrict__ freelist, int32_t* __restrict__ fs)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const bool is_free = c < nclasses && cstamp[c] < min_live;
    const unsigned long long m = __ballot(is_free);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(&fs[0], __popcll(m));
    base = __shfl(base, 0);
    if (is_free) freelist[base + __popcll(m & ((1ull << lane) - 1ull))] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&fs[1], 1) == (int)gridDim.x - 1) {   // the last workgroup: the list is complete
            fs[0] = 0;
            fs[1] = 0;
        }
    }
}

// ---- a received record (sharded sessions) -> staging row n + p of the means and a class of its own
__global__ __launch_bounds__(256) void migrate_unpack_split_kernel(const float* __restrict__ in, int total, int n, float* __restrict__ pose,
                                                                   int64_t pose_ld, float* __restrict__ mean, float* __restrict__ cov,
                                                                   float* __restrict__ covx, int32_t* __restrict__ cls, int Lp,
                                                                   int nlandmarks, float q, const int32_t* __restrict__ freelist,
                                                                   int first, uint32_t* __restrict__ cstamp, uint32_t stamp,
                                                                   int32_t* __restrict__ live, int32_t* __restrict__ cnt)
{
    const int p = blockIdx.x;
    if (p >= total) return;
    const float* __restrict__ rec = in + (int64_t)(3 + 5 * nlandmarks) * p;
    const int c = freelist[first + p];
    if (threadIdx.x < 3) pose[threadIdx.x * pose_ld + n + p] = rec[threadIdx.x];
    float* m = mean + (int64_t)(n + p) * 2 * Lp;
    float* cr = cov + (int64_t)c * 3 * Lp;
    float* xr = covx + (int64_t)c * 2 * Lp;
    for (int l = threadIdx.x; l < Lp; l += 256) {
        const bool in_row = l < nlandmarks;
        m[l] = in_row ? rec[3 + l] : 0.0f;
        m[Lp + l] = in_row ? rec[3 + nlandmarks + l] : 0.0f;
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
        cstamp[c] = stamp;
        live[atomicAdd(cnt, 1)] = c;
    }
}

__global__ __launch_bounds__(256) void class_gather_kernel(const int32_t* __restrict__ cls_in, int32_t* __restrict__ cls_out,
                                                           const int32_t* __restrict__ anc, int n, uint32_t* __restrict__ cstamp,
                                                           uint32_t stamp_now)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = cls_in[anc ? anc[i] : i];
    cls_out[i] = c;
    cstamp[c] = stamp_now;
}

// ---- reset: every landmark of every particle "not seen yet" (P_xx = -1), one class
__global__ __launch_bounds__(256) void split_reset_kernel(float* __restrict__ mean, float* __restrict__ cov, int32_t* __restrict__ cls,
                                                          int Lp, int n, int32_t* __restrict__ live, int32_t* __restrict__ cnt, int phase,
                                                          uint32_t* __restrict__ cstamp, uint32_t stamp_now, int32_t* __restrict__ h_live,
                                                          uint32_t epoch)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < (int64_t)n * 2 * Lp) mean[idx] = 0.0f;
    if (idx < 3 * (int64_t)Lp) cov[idx] = idx < Lp ? -1.0f : 0.0f;
    if (idx < n) cls[idx] = 0;
    if (idx == 0) {
        live[0] = 0;
        cstamp[0] = stamp_now;
        cnt[phase] = 1;
        cnt[(phase + 1) % 3] = 0;
        cnt[(phase + 2) % 3] = 0;
        if (h_live) publish_live(h_live, 1, epoch);
    }
}

}  // namespace

hipError_t launch_cov_update(hipStream_t stream, const CovArgs& a, int bound, const EventPair* ev)
{
    if (bound <= 0) return hipSuccess;
    if (ev) (void)hipEventRecord(ev->start, stream);
    // nlandmarks == 0: a frame without observations — only the list is brought up to date (classes whose last particle went
    // with the frame's gather leave it: a sharded session may hand their numbers out again, and a number must not be listed twice)
    cov_update_kernel<<<dim3((unsigned)bound, (unsigned)(a.nlandmarks > 0 ? (a.nlandmarks + 255) / 256 : 1)), 256, 0, stream>>>(a);
    if (ev) (void)hipEventRecord(ev->stop, stream);
    return hipGetLastError();
}

// flags u64[n] | sums u64[n] | scratch of the prefix sum
size_t split_scratch_words(int n) { return 2 * (2 * (size_t)n + (size_t)prefix_sum_scratch_elems(n)) + 4; }

hipError_t launch_split_from_rows(hipStream_t stream, const float* rows, int64_t row_stride_in, int plane_stride_in, int nlandmarks,
                                  int n, int Lp, float* mean, float* cov, float* covx, float meas_var, int32_t* cls, int32_t* live,
                                  int32_t* cnt, int phase, uint32_t* cstamp, uint32_t stamp_now, int32_t* h_live, uint32_t epoch,
                                  void* scratch)
{
    if (n <= 0) return hipSuccess;
    uint64_t* flag = static_cast<uint64_t*>(scratch);
    uint64_t* sum = flag + n;
    uint64_t* ps = sum + n;
    cov_heads_kernel<<<n, 256, 0, stream>>>(rows, row_stride_in, plane_stride_in, nlandmarks, n, flag);
    hipError_t err = launch_prefix_sum(stream, flag, n, sum, ps);
    if (err != hipSuccess) return err;
    split_from_rows_kernel<<<n, 256, 0, stream>>>(rows, row_stride_in, plane_stride_in, nlandmarks, n, Lp, flag, sum, mean, cov, cls,
                                                  live, cnt, phase, cstamp, stamp_now, h_live, epoch);
    // at most n classes: the launch is as wide as that, workgroups beyond the count leave at once
    cov_terms_kernel<<<dim3((unsigned)n, (unsigned)((Lp + 255) / 256)), 256, 0, stream>>>(cov, covx, Lp, nlandmarks, meas_var, cnt + phase);
    return hipGetLastError();
}

hipError_t launch_rows_from_split(hipStream_t stream, const float* mean, const float* cov, const int32_t* cls, int Lp,
                                  const int32_t* idx, int count, float* rows, int64_t row_stride, int plane_stride, int nlandmarks)
{
    if (count <= 0) return hipSuccess;
    rows_from_split_kernel<<<count, 256, 0, stream>>>(mean, cov, cls, Lp, idx, count, rows, row_stride, plane_stride, nlandmarks);
    return hipGetLastError();
}

hipError_t launch_class_free_list(hipStream_t stream, const uint32_t* cstamp, int nclasses, uint32_t min_live, int32_t* freelist,
                                  int32_t* fs)
{
    class_free_list_kernel<<<blocks256(nclasses), 256, 0, stream>>>(cstamp, nclasses, min_live, freelist, fs);
    return hipGetLastError();
}

hipError_t launch_migrate_unpack_split(hipStream_t stream, const float* in, int total, int n, float* pose, int64_t pose_ld, float* mean,
                                       float* cov, float* covx, int32_t* cls, int Lp, int nlandmarks, float meas_var,
                                       const int32_t* freelist, int first, uint32_t* cstamp, uint32_t stamp, int32_t* live,
                                       int32_t* cnt)
{
    if (total <= 0) return hipSuccess;
    migrate_unpack_split_kernel<<<total, 256, 0, stream>>>(in, total, n, pose, pose_ld, mean, cov, covx, cls, Lp, nlandmarks, meas_var,
                                                          freelist, first, cstamp, stamp, live, cnt);
    return hipGetLastError();
}

hipError_t launch_split_gather(hipStream_t stream, const float* mean_in, float* mean_out, const int32_t* cls_in, int32_t* cls_out,
                               int Lp, const int32_t* anc, int n, uint32_t* cstamp, uint32_t stamp_now)
{
    if (n <= 0) return hipSuccess;
    split_gather_kernel<<<n, 256, 0, stream>>>(mean_in, mean_out, cls_in, cls_out, Lp, anc, n, cstamp, stamp_now);
    return hipGetLastError();
}

hipError_t launch_class_gather(hipStream_t stream, const int32_t* cls_in, int32_t* cls_out, const int32_t* anc, int n, uint32_t* cstamp,
                               uint32_t stamp_now)
{
    if (n <= 0) return hipSuccess;
    class_gather_kernel<<<blocks256(n), 256, 0, stream>>>(cls_in, cls_out, anc, n, cstamp, stamp_now);
    return hipGetLastError();
}

hipError_t launch_split_reset(hipStream_t stream, float* mean, float* cov, float* covx, int32_t* cls, int Lp, int n, int32_t* live,
                              int32_t* cnt, int phase, uint32_t* cstamp, uint32_t stamp_now, int32_t* h_live, uint32_t epoch)
{
    const int64_t m = (int64_t)n * 2 * Lp;
    split_reset_kernel<<<blocks256(m > 3 * (int64_t)Lp ? m : 3 * (int64_t)Lp), 256, 0, stream>>>(mean, cov, cls, Lp, n, live, cnt, phase,
                                                                                              cstamp, stamp_now, h_live, epoch);
    cov_terms_kernel<<<dim3(1, (unsigned)((Lp + 255) / 256)), 256, 0, stream>>>(cov, covx, Lp, 0, 1.0f, cnt + phase);   // nothing seen yet
    return hipGetLastError();
}

}  // namespace slam
