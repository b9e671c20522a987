// The covariance classes' update of the split layout (split_kernels.hip: cov_update_kernel), as a body that a launch of
// another kernel can carry in workgroups of its own: the classes' update depends on the landmark update's stamps only and
// nothing behind it in the frame depends on it, so it travels in the launch of the weights (pf_kernels.hip:
// logweight_kernel) instead of costing a launch of its own (4-5 us of a 0.12 ms frame).
// No counterpart in the reference (it has no particles or landmarks).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ekf_math.h"
#include "kernels.h"

namespace slam {

// {classes in use, epoch} to mapped host memory as ONE 8-byte store (8-byte aligned), so that the host never pairs the count of
// one epoch with the number of another
__device__ __forceinline__ void publish_live(int32_t* h_live, int count, uint32_t epoch)
{
    *reinterpret_cast<volatile unsigned long long*>(h_live) = (unsigned long long)(uint32_t)count | ((unsigned long long)epoch << 32);
}

// workgroup (k, y) of 256 threads takes landmarks [256 y, 256 y + 256) of class live[k]
__device__ __forceinline__ void cov_update_body(const CovArgs& a, int k, int y)
{
    const int nlive = a.cnt[a.phase];
    if (k == 0 && y == 0 && threadIdx.x == 0) {
        a.cnt[(a.phase + 2) % 3] = 0;   // the list after next: nobody reads or writes it during this launch
        if (a.h_live) {
            publish_live(a.h_live, nlive, a.epoch);
            __threadfence_system();   // the count first: a host that pairs a newer count with an older mark over-estimates
            publish_live(a.h_mark, (int)a.mark, a.epoch);
        }
    }
    if (k >= nlive) return;
    const int c = a.live_in[k];
    if (a.cstamp[c] != a.stamp_now) return;   // its last particle is gone: the class leaves the list
    if (y == 0 && threadIdx.x == 0) a.live_out[atomicAdd(&a.cnt[(a.phase + 1) % 3], 1)] = c;
    const int l = y * 256 + threadIdx.x;
    if (l >= a.nlandmarks) return;   // (the padding of a row is never observed: it stays as it is)
    const float zx = a.obs_zx[l], zy = a.obs_zy[l];
    if (!(zx == zx && zy == zy)) return;   // not observed: the prior stays
    float* row = a.cov + (int64_t)c * a.cov_stride + l;
    const float pxx = row[0], pxy = row[a.plane_stride], pyy = row[2 * (int64_t)a.plane_stride];
    float o2 = a.meas_var, o3 = 0.0f, o4 = a.meas_var;   // a first sighting: q I
    if (!(pxx < 0.0f)) {
        // (the prior's determinant terms lie in covx; starting from them instead of recomputing them gives the same bits)
        const float* xr = a.covx + (int64_t)c * a.covx_stride + l;
        const EkfShared<float> h = ekf_shared_from<float>(pxx, pxy, pyy, a.meas_var, xr[0], xr[a.plane_stride]);
        o2 = h.o2;
        o3 = h.o3;
        o4 = h.o4;
    }
    row[0] = o2;
    row[a.plane_stride] = o3;
    row[2 * (int64_t)a.plane_stride] = o4;
    // what the next update of this landmark starts from
    float idet, hl;
    ekf_det_terms<float>(o2, o3, o4, a.meas_var, idet, hl);
    float* xw = a.covx + (int64_t)c * a.covx_stride + l;
    xw[0] = idet;
    xw[a.plane_stride] = hl;
}

}  // namespace slam
