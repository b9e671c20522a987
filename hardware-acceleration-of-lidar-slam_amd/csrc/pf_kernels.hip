// pf_kernels.hip — particle-filter stages around the scan matcher.
//
// None of these stages exists in the reference: the specification is DESIGN.md
// + oracle/slam_oracle_pf.c, and these kernels match that specification bit for bit.  The only
// reference anchor is the zero-noise motion step = the constant-velocity predict of
// Subsystem_1/main.c:875-898.
//
// Data layout (HBM): particles are SoA float arrays; the landmark maps are ONE ROW PER PARTICLE,
// [particle][5 planes: mu_x, mu_y, P_xx, P_xy, P_yy][plane_stride floats], so that a wavefront walking one
// particle's landmarks moves 256 contiguous bytes per plane and access, and the offspring of one resample
// ancestor (neighbouring particles) share its row through L2.
// All kernels are HBM-streaming or latency-bound integer work; there is no GEMM shape here
// (the largest matrix is 2x2), hence no MFMA.

#include <stdlib.h>

#include "det_math.h"
#include "ekf_math.h"
#include "score_body.h"
#include "cov_update_body.h"
#include "kernels.h"

namespace slam {

namespace {

constexpr int kBlock = 256;

// ------------------------------------------------------------------ A9: motion sample
__global__ __launch_bounds__(kBlock) void motion_sample_kernel(const float* __restrict__ sx,
                                                               const float* __restrict__ sy,
                                                               const float* __restrict__ sth,
                                                               const int32_t* __restrict__ anc, float* __restrict__ x,
                                                               float* __restrict__ y, float* __restrict__ th, int n,
                                                               MotionParams mp)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int j = anc ? anc[i] : i;
    float ox, oy, ot;
    motion_sample_one(mp, (uint64_t)i, sx[j], sy[j], sth[j], ox, oy, ot);
    x[i] = ox;
    y[i] = oy;
    th[i] = ot;
}

// ------------------------------------------------------------------ A10: 2x2 EKF per (particle, landmark)
// The map is one row per particle (5 planes of plane_stride floats).  ONE WAVEFRONT OWNS ONE PARTICLE and its
// lanes walk the landmarks of the row, two landmarks per lane (l and l + 64 of each batch of 128), so that every
// load and store is a coalesced 256-byte access and the arithmetic runs on float2 (v_pk_mul_f32 / v_pk_add_f32:
// IEEE per component, i.e. the same bits as the scalar form).  The observations of the frame come as a table indexed by
// landmark (zx[l], zy[l], NaN = not observed), read alongside the row.  Why rows: after a resample most
// particles are copies of few ancestors (the bench's filter keeps ~6 % distinct), the offspring of one ancestor
// are neighbouring particles, so the 10 KB source row is fetched from HBM once and re-read from L2 by the other
// offspring — the sweep's HBM traffic is the 20 B/(particle, landmark) it writes plus the distinct rows it reads,
// not 40 B.  Row base addresses are wave-uniform (SGPR).
__device__ __forceinline__ float wave_xor_tree_sum(float v)   // t[j] = t[j] + t[j ^ s], s = 1 .. 32: all lanes equal
{
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) v = v + __shfl_xor(v, s, 64);
    return v;
}

constexpr int kEkfWaves = 4;   // particles per workgroup

// global-address-space pointers: "scalar base + 32-bit lane offset" is an addressing mode of global_load/store only
typedef __attribute__((address_space(1))) char gchar;
typedef __attribute__((address_space(1))) float gfloat;
// cache policy of the row stores: 2 = nt (streaming; the written rows are next read a frame later, long after they
// left the caches).  Measured at 64k x 500: default 170 us, nt 162 us, sc0 170 us, sc1 171 us in the filter;
// 243 / 248 / 244 / 243 us for a sweep without shared ancestors.
#ifndef EKF_STORE_AUX
#define EKF_STORE_AUX 2
#endif
__device__ __forceinline__ float row_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
}
__device__ __forceinline__ void row_store(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff, float v)
{
#ifdef EKF_MEASURE_NO_STORES   // measurement builds only (wrong results): what the kernels take without their row stores
    if (__float_as_uint(v) == 0x7fc12345u)
#endif
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)voff, soff, EKF_STORE_AUX);
}
__device__ __forceinline__ gchar* uniform_gptr(const void* p)   // tell the compiler the pointer is wave-uniform
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (gchar*)(((uint64_t)hi << 32) | lo);
}

struct EkfLane {   // per-wavefront constants of one particle
    // source row and destination row as buffer resources (wave-uniform descriptors in SGPRs): an access is
    // "descriptor + 32-bit lane offset + scalar plane offset", no 64-bit vector arithmetic for loads or stores
    __amdgpu_buffer_rsrc_t rin, rout;
    int pl;   // plane stride in bytes
    const gchar *ozx, *ozy;
    unsigned L;
    v2f s, c, px, py, q;
};

// What goes into the row for the two landmarks of a lane, given the update's result in r0 .. r4 / ll: a first sighting
// (prior P_xx < 0) takes the observed point and P = q I and adds no likelihood term; a landmark without an observation keeps
// its prior values.  Both cases are decided for the WAVEFRONT first (a ballot each): in a running filter most batches of
// 128 landmarks hold neither — every landmark seen before, every one observed, or none — and then the selects (and the
// arithmetic of the first sighting) are skipped altogether.  The values are those of
//     ob ? (first ? {f0, f1, q, 0, q; 0} : {o0 .. o4; ll}) : {prior; 0}
// in every case.
__device__ __forceinline__ void ekf_select(v2f& r0, v2f& r1, v2f& r2, v2f& r3, v2f& r4, v2f& ll, v2f mx, v2f my, v2f pxx, v2f pxy,
                                           v2f pyy, v2f zx, v2f zy, v2f s, v2f c, v2f px, v2f py, v2f q, bool ob0, bool ob1)
{
    if (__ballot(pxx[0] < 0.0f || pxx[1] < 0.0f) != 0) {
        v2f f0, f1;
        ekf_first_sighting<v2f>(zx, zy, s, c, px, py, f0, f1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bool first = pxx[t] < 0.0f;
            r0[t] = first ? f0[t] : r0[t];
            r1[t] = first ? f1[t] : r1[t];
            r2[t] = first ? q[t] : r2[t];
            r3[t] = first ? 0.0f : r3[t];
            r4[t] = first ? q[t] : r4[t];
            ll[t] = first ? 0.0f : ll[t];
        }
    }
    if (__ballot(!(ob0 && ob1)) != 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bool ob = t ? ob1 : ob0;
            r0[t] = ob ? r0[t] : mx[t];
            r1[t] = ob ? r1[t] : my[t];
            r2[t] = ob ? r2[t] : pxx[t];
            r3[t] = ob ? r3[t] : pxy[t];
            r4[t] = ob ? r4[t] : pyy[t];
            ll[t] = ob ? ll[t] : 0.0f;
        }
    }
}

// NB batches of 128 landmarks starting at lb: all loads first, then the arithmetic, then the stores.  A lane owns
// landmarks l and l + 64 of each batch, so every access is one 256-byte dword access per wavefront (8-byte
// accesses, a lane owning neighbours, were measured ~20 % slower whenever the source rows come out of L2).
// FULL: every lane's landmarks lie inside the row (lb + 128*NB <= plane_stride) and the update is out of place,
// so nothing is predicated; landmarks at or beyond L (row padding) then simply count as "not observed" and their
// padding values are copied along.  !FULL: the general form (row tails, in-place updates).
template <int NB, bool FULL, bool COPY>
__device__ __forceinline__ void ekf_batches(const EkfLane& w, unsigned lb, unsigned lane, v2f& acc)
{
    const float nan = __uint_as_float(0x7fc00000u);
    v2f m[NB][5], zx[NB], zy[NB];
    unsigned off[NB][2];
    bool obs[NB][2], use[NB][2];
#pragma unroll
    for (int g = 0; g < NB; ++g)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned l = lb + (unsigned)g * 128u + 64u * t + lane;
            const bool in = l < w.L;
            off[g][t] = ((FULL || in) ? l : 0u) * 4u;
            const unsigned zo = (in ? l : 0u) * 4u;   // clamped index + select instead of a predicated load
            const float vx = *(const gfloat*)(w.ozx + zo), vy = *(const gfloat*)(w.ozy + zo);
            zx[g][t] = in ? vx : nan;
            zy[g][t] = in ? vy : nan;
            // NaN = no observation (also what lanes beyond L were given).  Testing zy as well keeps its load up here
            // with the others: the compiler otherwise sinks it into the arithmetic, two extra round trips per batch.
            obs[g][t] = zx[g][t] == zx[g][t] && zy[g][t] == zy[g][t];
            use[g][t] = FULL ? true : (COPY ? in : obs[g][t]);
        }
#pragma unroll
    for (int g = 0; g < NB; ++g)
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (FULL || use[g][t]) {
#pragma unroll
                for (int p = 0; p < 5; ++p) m[g][p][t] = row_load(w.rin, off[g][t], p * w.pl);
            }
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        if (!FULL && !(use[g][0] || use[g][1])) continue;
        const v2f mx = m[g][0], my = m[g][1], pxx = m[g][2], pxy = m[g][3], pyy = m[g][4];
        if (COPY && __ballot(obs[g][0] || obs[g][1]) == 0) {   // no observation among these 128 landmarks: plain copy
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (FULL || use[g][t]) {
#pragma unroll
                    for (int p = 0; p < 5; ++p) row_store(w.rout, off[g][t], p * w.pl, m[g][p][t]);
                }
            continue;
        }
        const v2f q = w.q;
        const EkfResult<v2f> u = ekf_update_one<v2f, false>(mx, my, pxx, pxy, pyy, zx[g], zy[g], w.s, w.c, w.px, w.py, q);
        v2f r0 = u.o0, r1 = u.o1, r2 = u.o2, r3 = u.o3, r4 = u.o4, ll = u.ll;
        ekf_select(r0, r1, r2, r3, r4, ll, mx, my, pxx, pxy, pyy, zx[g], zy[g], w.s, w.c, w.px, w.py, q, obs[g][0], obs[g][1]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (FULL || use[g][t]) {
                row_store(w.rout, off[g][t], 0 * w.pl, r0[t]);
                row_store(w.rout, off[g][t], 1 * w.pl, r1[t]);
                row_store(w.rout, off[g][t], 2 * w.pl, r2[t]);
                row_store(w.rout, off[g][t], 3 * w.pl, r3[t]);
                row_store(w.rout, off[g][t], 4 * w.pl, r4[t]);
            }
        acc = acc + ll;
    }
}

// NB: batches of 128 landmarks per pass of the fast path.  COPY: out of place.
template <int NB, bool COPY>
__global__ __launch_bounds__(kEkfWaves * 64) void ekf_update_kernel(EkfArgs a)
{
    const unsigned lane = threadIdx.x & 63u;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Workgroups are dealt to the 8 XCDs round-robin.  Renumber them so that each XCD (one L2) works on one
    // contiguous eighth of the particles: the offspring of an ancestor then share ONE L2 instead of up to eight.
    int bid = blockIdx.x;
    if (a.xcd_chunk > 0) {
        const int per = a.xcd_chunk;   // workgroups per XCD, gridDim.x == 8 * per
        bid = (bid & 7) * per + (bid >> 3);
    }
    const int i = bid * kEkfWaves + wave;
    if (i >= a.n) return;
    const int src = a.anc ? a.anc[i] : i;
    float st_, ct_;
    det_sincosf(a.th[i], st_, ct_);
    EkfLane w;
    const int row_bytes = __builtin_amdgcn_readfirstlane(5 * a.plane_stride * 4);
    w.rin = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_gptr(a.map_in + (int64_t)src * a.row_stride), 0, row_bytes, 0x00020000);
    w.rout = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_gptr(a.map_out + (int64_t)i * a.row_stride), 0, row_bytes, 0x00020000);
    w.pl = __builtin_amdgcn_readfirstlane(a.plane_stride * 4);
    w.ozx = uniform_gptr(a.obs_zx);
    w.ozy = uniform_gptr(a.obs_zy);
    w.L = (unsigned)a.nlandmarks;
    w.s = bc2(st_); w.c = bc2(ct_); w.px = bc2(a.x[i]); w.py = bc2(a.y[i]); w.q = bc2(a.meas_var);

    v2f acc = bc2(0.0f);   // lane j: .x = accumulator j, .y = accumulator j + 64 of the spec (landmark l -> l mod 128)
    unsigned lb = 0;
    if (COPY) {   // whole batches that fit into the row, padding included: nothing predicated
        const unsigned room = (unsigned)a.plane_stride;
        for (; lb < w.L && lb + 128u * NB <= room; lb += 128u * NB) ekf_batches<NB, true, COPY>(w, lb, lane, acc);
        if (NB > 1)
            for (; lb < w.L && lb + 128u <= room; lb += 128u) ekf_batches<1, true, COPY>(w, lb, lane, acc);
    }
    // the general form (row tails; in-place updates).  In place only observed landmarks are touched, so a wavefront's
    // time is round trips, not bytes: NB batches go through one round trip together (batches beyond L load nothing).
    constexpr int NBT = COPY ? 1 : NB;
    for (; lb < w.L; lb += 128u * NBT) ekf_batches<NBT, false, COPY>(w, lb, lane, acc);

    const float total = wave_xor_tree_sum(acc[0] + acc[1]);
    if (lane == 0) {
        a.loglik[i] = total;
        if (a.loglik_user) a.loglik_user[i] = total;
    }
}

// ---- grouped form of the out-of-place update: ONE WAVEFRONT OWNS G NEIGHBOURING PARTICLES (G = 2, 4 or 8).
// After a resample the slots are sorted by ancestor, so neighbouring particles mostly descend from the same one.  The
// row-per-wavefront kernel lets them share the source row through L2; measured (profiles/copy_ceiling.hip) even a pure copy
// pays for that — 155 us at 64k x 512 columns when 16 neighbours share a source, against 109 us when nothing is re-read.
// Here the wavefront walks the landmarks in the OUTER loop and its G particles in the inner one: a batch of the source
// row stays in registers while every particle of the group that descends from it is updated with its own pose and stored
// to its own row — the re-reads never leave the register file, the observation table is read once per group.  Per
// (particle, landmark) the arithmetic, its order and the log-likelihood summation are those of ekf_batches (bit-exact:
// the same tests cover both kernels); the per-particle accumulators live in LDS between batches.
template <int NB>
struct EkfBatch {   // NB batches of 128 landmarks of one source row + the observations of those landmarks
    v2f mx[NB], my[NB];        // prior means
    v2f p2[NB], p3[NB], p4[NB];   // what goes into the covariance planes: (I - W) P, q I for a first sighting, the prior without an observation
    EkfShared<v2f> sh[NB];     // the pose-independent part of the update (csrc/ekf_math.h), worked out once per source row
    v2f zx[NB], zy[NB];
    bool obs[NB][2], first[NB][2];
    bool any_obs[NB], all_obs[NB], any_first[NB];   // wave-uniform
    unsigned off[NB][2];
};

struct EkfPose {   // one particle of the group (wave-uniform values)
    __amdgpu_buffer_rsrc_t rout;
    v2f s, c, px, py;
};

// A new source row is in registers (b.mx / b.my and the prior covariance pxx / pxy / pyy of batch g): everything about it
// that does not depend on the particle — the gain, the posterior covariance, the determinant's logarithm (ekf_shared) and
// the selection of what the covariance planes receive (a first sighting: q I; no observation: the prior).
template <int NB>
__device__ __forceinline__ void ekf_prepare(EkfBatch<NB>& b, int g, v2f pxx, v2f pxy, v2f pyy, v2f q)
{
    b.p2[g] = pxx;
    b.p3[g] = pxy;
    b.p4[g] = pyy;
    b.first[g][0] = pxx[0] < 0.0f;
    b.first[g][1] = pxx[1] < 0.0f;
    b.any_first[g] = __ballot(b.first[g][0] || b.first[g][1]) != 0;
    if (!b.any_obs[g]) return;   // no observation among these 128 landmarks: the rows are copied
    b.sh[g] = ekf_shared<v2f>(pxx, pxy, pyy, q);
    v2f r2 = b.sh[g].o2, r3 = b.sh[g].o3, r4 = b.sh[g].o4;
    if (b.any_first[g]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            r2[t] = b.first[g][t] ? q[t] : r2[t];
            r3[t] = b.first[g][t] ? 0.0f : r3[t];
            r4[t] = b.first[g][t] ? q[t] : r4[t];
        }
    }
    if (!b.all_obs[g]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            r2[t] = b.obs[g][t] ? r2[t] : pxx[t];
            r3[t] = b.obs[g][t] ? r3[t] : pxy[t];
            r4[t] = b.obs[g][t] ? r4[t] : pyy[t];
        }
    }
    b.p2[g] = r2;
    b.p3[g] = r3;
    b.p4[g] = r4;
}

// update NB prepared batches with one particle's pose and store them to its row (FULL batches only: every lane's landmarks
// lie inside the padded row; landmarks beyond L count as "not observed", padding is copied along).  Per particle there is
// the observed point in the world frame, the innovation, the new mean and the likelihood term (ekf_particle); the values
// are those of ekf_update_one + ekf_select.
template <int NB>
__device__ __forceinline__ void ekf_apply(const EkfBatch<NB>& b, const EkfPose& w, int pl, v2f& acc)
{
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        v2f r0 = b.mx[g], r1 = b.my[g];
        if (b.any_obs[g]) {
            const EkfParticle<v2f> u = ekf_particle<v2f>(b.sh[g], b.mx[g], b.my[g], b.zx[g], b.zy[g], w.s, w.c, w.px, w.py);
            v2f ll = u.ll;
            r0 = u.o0;
            r1 = u.o1;
            if (b.any_first[g]) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    r0[t] = b.first[g][t] ? u.wx[t] : r0[t];
                    r1[t] = b.first[g][t] ? u.wy[t] : r1[t];
                    ll[t] = b.first[g][t] ? 0.0f : ll[t];
                }
            }
            if (!b.all_obs[g]) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    r0[t] = b.obs[g][t] ? r0[t] : b.mx[g][t];
                    r1[t] = b.obs[g][t] ? r1[t] : b.my[g][t];
                    ll[t] = b.obs[g][t] ? ll[t] : 0.0f;
                }
            }
            acc = acc + ll;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            row_store(w.rout, b.off[g][t], 0 * pl, r0[t]);
            row_store(w.rout, b.off[g][t], 1 * pl, r1[t]);
            row_store(w.rout, b.off[g][t], 2 * pl, b.p2[g][t]);
            row_store(w.rout, b.off[g][t], 3 * pl, b.p3[g][t]);
            row_store(w.rout, b.off[g][t], 4 * pl, b.p4[g][t]);
        }
    }
}


__device__ __forceinline__ float lane_value(float v, int k)   // lane k's value, wave-uniform (v_readlane_b32)
{
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), k));
}

// EKF_GROUP_NB: batches of 128 landmarks a wavefront of the grouped kernels holds in registers per pass; EKF_GROUP_WPE: waves
// per SIMD the register allocation is held to.  With the pose-independent part of the update hoisted (ekf_prepare) a batch
// costs 13 register pairs: two batches need 97 VGPRs (5 waves at 96 with two dwords of scratch), one batch 69 (7 waves).
// Interleaved A/B on one box (profiles/ab.py, 64k x 500 in the filter): 2 batches at 5 waves — fused front 125.7 us
// (0.1558 ms per frame), update alone 122-134 us; 1 batch at 7 waves — fused front 143.8 us (0.1678 ms), update alone
// 125 us; 1M x 1000: 4.23 against 4.31 ms fused, 4.19 against 4.30 ms alone.  2 batches at 6 waves spill 13 dwords
// (161-178 us), 1 batch at 8 waves 6 dwords (151-167 us).  Before the hoisting (sensor-frame arithmetic, 80 VGPRs, 2 batches
// at 6 waves): fused front 148-153 us, update alone 133-149 us.
// 4 waves (97 VGPRs, nothing spilled) against 5 on another, slower box: fused front 142.7 against 147.3 us, update alone
// 145.0 against 146.3 us; equal at 2000 landmarks and with 32 of 500 observed.
#ifndef EKF_GROUP_WPE
#define EKF_GROUP_WPE 4
#endif
#ifndef EKF_GROUP_NB
#define EKF_GROUP_NB 2
#endif
#define EKF_GROUP_ATTR __attribute__((amdgpu_waves_per_eu(EKF_GROUP_WPE, EKF_GROUP_WPE)))
// the same two knobs for the kernels of the split layout (a batch costs fewer registers there: no covariance planes to carry)
#ifndef EKF_SPLIT_WPE
#define EKF_SPLIT_WPE 5   // 64k x 500, fused front: 4 waves 97.7 us, 5 waves 94.8 us, 6 waves 99.4 us, 8 waves (spills) 149 us
#endif
#ifndef EKF_SPLIT_NB
#define EKF_SPLIT_NB 2
#endif
#define EKF_SPLIT_ATTR __attribute__((amdgpu_waves_per_eu(EKF_SPLIT_WPE, EKF_SPLIT_WPE)))
#define EKF_FRONT_ATTR(SPLIT_) __attribute__((amdgpu_waves_per_eu((SPLIT_) ? EKF_SPLIT_WPE : EKF_GROUP_WPE, (SPLIT_) ? EKF_SPLIT_WPE : EKF_GROUP_WPE)))
// `bid`: the workgroup's index after the XCD-contiguous renumbering; s_acc: per particle of the group the 128 accumulators of
// the specification.  OWN_MOTION (the fused front kernel of a frame, below): the poses are not read from a.x / a.y / a.th but
// worked out here — pose = motion_sample(source pose of the ancestor), the very computation the scoring workgroups of the
// same launch make for the same particle (Philox is counter-based: the same bits) — so that the update waits for nobody.
template <int NB, int G, bool OWN_MOTION>
__device__ __forceinline__ void ekf_group_body(const EkfArgs& a, int bid, float (*s_acc)[G][128], const MotionIO& mio,
                                               const MotionParams& mpar)
{
    const unsigned lane = threadIdx.x & 63u;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g0 = (bid * kEkfWaves + wave) * G;
    if (g0 >= a.n) return;
    const int nslots = a.n - g0 < G ? a.n - g0 : G;
    // lane k prepares particle g0 + k: its source row and the trig of its heading; read back with v_readlane below
    const int mine = g0 + ((int)lane < nslots ? (int)lane : 0);
    const int src_l = a.anc ? a.anc[mine] : mine;
    float st_l, ct_l, px_l, py_l;
    if constexpr (OWN_MOTION) {
        float th_l;
        motion_sample_one(mpar, (uint64_t)mine, mio.sx[src_l], mio.sy[src_l], mio.sth[src_l], px_l, py_l, th_l);
        det_sincosf(th_l, st_l, ct_l);
    } else {
        det_sincosf(a.th[mine], st_l, ct_l);
        px_l = a.x[mine];
        py_l = a.y[mine];
    }
#pragma unroll
    for (int k = 0; k < G; ++k) {
        s_acc[wave][k][lane] = 0.0f;
        s_acc[wave][k][lane + 64] = 0.0f;
    }
    const int pl = __builtin_amdgcn_readfirstlane(a.plane_stride * 4);
    const int row_bytes = __builtin_amdgcn_readfirstlane(5 * a.plane_stride * 4);
    const gchar* ozx = uniform_gptr(a.obs_zx);
    const gchar* ozy = uniform_gptr(a.obs_zy);
    const unsigned L = (unsigned)a.nlandmarks, room = (unsigned)a.plane_stride;
    const v2f q2 = bc2(a.meas_var);
    const float nan = __uint_as_float(0x7fc00000u);

    auto pose_of = [&](int k) {
        EkfPose w;
        const int i = g0 + k;
        w.rout = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_gptr(a.map_out + (int64_t)i * a.row_stride), 0, row_bytes, 0x00020000);
        w.s = bc2(lane_value(st_l, k));
        w.c = bc2(lane_value(ct_l, k));
        w.px = bc2(lane_value(px_l, k));
        w.py = bc2(lane_value(py_l, k));
        return w;
    };

    unsigned lb = 0;
    for (; lb < L && lb + 128u * NB <= room; lb += 128u * NB) {
        EkfBatch<NB> b;
        // the observations of these landmarks: the same for every particle of the group
#pragma unroll
        for (int g = 0; g < NB; ++g)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const unsigned l = lb + (unsigned)g * 128u + 64u * t + lane;
                const bool in = l < L;
                b.off[g][t] = l * 4u;
                const unsigned zo = (in ? l : 0u) * 4u;   // clamped index + select instead of a predicated load
                const float vx = *(const gfloat*)(ozx + zo), vy = *(const gfloat*)(ozy + zo);
                b.zx[g][t] = in ? vx : nan;
                b.zy[g][t] = in ? vy : nan;
                b.obs[g][t] = b.zx[g][t] == b.zx[g][t] && b.zy[g][t] == b.zy[g][t];
            }
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            b.any_obs[g] = __ballot(b.obs[g][0] || b.obs[g][1]) != 0;
            b.all_obs[g] = __ballot(!(b.obs[g][0] && b.obs[g][1])) == 0;
        }
        int prev = -1;
        for (int k = 0; k < nslots; ++k) {
            const int src = __builtin_amdgcn_readlane(src_l, k);
            if (src != prev) {   // a new ancestor: its batch into registers (wave-uniform branch)
                const __amdgpu_buffer_rsrc_t rin =
                    __builtin_amdgcn_make_buffer_rsrc((void*)uniform_gptr(a.map_in + (int64_t)src * a.row_stride), 0, row_bytes, 0x00020000);
                v2f pr[NB][3];
#pragma unroll
                for (int g = 0; g < NB; ++g)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        b.mx[g][t] = row_load(rin, b.off[g][t], 0 * pl);
                        b.my[g][t] = row_load(rin, b.off[g][t], 1 * pl);
#pragma unroll
                        for (int p = 0; p < 3; ++p) pr[g][p][t] = row_load(rin, b.off[g][t], (2 + p) * pl);
                    }
#pragma unroll
                for (int g = 0; g < NB; ++g) ekf_prepare<NB>(b, g, pr[g][0], pr[g][1], pr[g][2], q2);
                prev = src;
            }
            const EkfPose w = pose_of(k);
            v2f acc = (v2f){s_acc[wave][k][lane], s_acc[wave][k][lane + 64]};
            ekf_apply<NB>(b, w, pl, acc);
            s_acc[wave][k][lane] = acc[0];
            s_acc[wave][k][lane + 64] = acc[1];
        }
    }
    // what is left of the rows (a tail shorter than NB batches) and the reduction: particle by particle, general form
    for (int k = 0; k < nslots; ++k) {
        const int i = g0 + k;
        const int src = __builtin_amdgcn_readlane(src_l, k);
        EkfLane w;
        w.rin = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_gptr(a.map_in + (int64_t)src * a.row_stride), 0, row_bytes, 0x00020000);
        const EkfPose pw = pose_of(k);
        w.rout = pw.rout;
        w.pl = pl;
        w.ozx = ozx;
        w.ozy = ozy;
        w.L = L;
        w.s = pw.s; w.c = pw.c; w.px = pw.px; w.py = pw.py; w.q = q2;
        v2f acc = (v2f){s_acc[wave][k][lane], s_acc[wave][k][lane + 64]};
        unsigned lt = lb;
        for (; lt < L && lt + 128u <= room; lt += 128u) ekf_batches<1, true, true>(w, lt, lane, acc);
        for (; lt < L; lt += 128u) ekf_batches<1, false, true>(w, lt, lane, acc);
        const float total = wave_xor_tree_sum(acc[0] + acc[1]);
        if (lane == 0) {
            a.loglik[i] = total;
            if (a.loglik_user) a.loglik_user[i] = total;
        }
    }
}

template <int NB, int G>
__global__ __launch_bounds__(kEkfWaves * 64) EKF_GROUP_ATTR void ekf_update_group_kernel(EkfArgs a)
{
    __shared__ float s_acc[kEkfWaves][G][128];
    int bid = blockIdx.x;
    if (a.xcd_chunk > 0) bid = (bid & 7) * a.xcd_chunk + (bid >> 3);   // each XCD a contiguous eighth (see ekf_update_kernel)
    ekf_group_body<NB, G, false>(a, bid, s_acc, MotionIO{}, MotionParams{});
}

// ---- the same grouped update on the SPLIT layout (EkfArgs::cov != nullptr): a particle's row holds its landmark MEANS only,
// the covariance planes exist once per covariance class (kernels.h; the classes' own update: split_kernels.hip).  Per particle
// and landmark the update then reads 8 bytes (the ancestor's means, kept in registers for the offspring in the group) and
// writes 8, instead of 20 and 20; the class's covariance row — the same few KB for every wavefront once the population
// descends from few classes — comes out of L2.  Arithmetic, operation order and log-likelihood summation are those of
// ekf_group_body (ekf_shared + ekf_particle): the same bits.  Rows are walked in whole passes of NB batches up to L; lanes
// whose landmarks lie beyond the row's planes get the buffer offset 0xffffffff, which the hardware's range check turns into
// "load 0, drop the store" (score_body.h uses the same device), so no pass needs a predicated form.
template <int NB>
struct SplitBatch {
    v2f mx[NB], my[NB];        // prior means of the current source row
    EkfShared<v2f> sh[NB];     // the pose-independent part of the update, from the current class's covariance row (o2 .. o4 unused)
    v2f zx[NB], zy[NB];
    // the two special cases of a landmark, as lane masks: `keep` = no observation (the prior mean stays, no likelihood term),
    // `first` = observed for the first time (the observed point becomes the mean, no likelihood term); wave-uniform: whether a
    // batch holds any observation at all, and whether it holds a special lane
    bool keep[NB][2], first[NB][2];
    bool any_obs[NB], any_keep[NB], special[NB];
    unsigned off[NB][2];
};

// One batch of one particle.  SPECIAL = false: every lane holds an observed landmark seen before — the plain update, no
// select anywhere.  In a running filter that is nearly every batch, and left to itself the compiler turns the two wave-uniform
// tests around the special cases into 26 v_cndmask per batch (as many instructions as the update's arithmetic: counted in
// the ISA of round 3's kernel): hence two copies of the batch, chosen by a REAL branch (the asm statement keeps the copies
// from being merged back into one).
template <int NB, bool SPECIAL>
__device__ __forceinline__ void split_apply_one(const SplitBatch<NB>& b, int g, const EkfPose& w, int pl, v2f& term)
{
    v2f zx = b.zx[g];
    if constexpr (SPECIAL) asm volatile("" : "+v"(zx));
    const EkfParticle<v2f> u = ekf_particle<v2f>(b.sh[g], b.mx[g], b.my[g], zx, b.zy[g], w.s, w.c, w.px, w.py);
    v2f r0 = u.o0, r1 = u.o1, ll = u.ll;
    if constexpr (!SPECIAL) {
        // The two landmarks of a lane are stored one by one, and left to itself the compiler pushes the two extracts up through
        // the whole expression and then packs each landmark's w00 * dx + w01 * dy as ONE product pair + a horizontal add — with
        // two register moves per pair to line the operands up: 20 instructions for the four new means where 8 packed ones do
        // (counted in the ISA, profiles/r04_split_tuning.md section 10).  The packed values are made opaque before the extracts.
        asm("" : "+v"(r0));
        asm("" : "+v"(r1));
    }
    if constexpr (SPECIAL) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {   // obs ? (first ? the observed point : the update) : the prior
            r0[t] = b.keep[g][t] ? b.mx[g][t] : (b.first[g][t] ? u.wx[t] : r0[t]);
            r1[t] = b.keep[g][t] ? b.my[g][t] : (b.first[g][t] ? u.wy[t] : r1[t]);
            ll[t] = (b.keep[g][t] || b.first[g][t]) ? 0.0f : ll[t];
        }
    }
    term = ll;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        row_store(w.rout, b.off[g][t], 0, r0[t]);
        row_store(w.rout, b.off[g][t], pl, r1[t]);
    }
}

// one particle, the NB batches of a pass: term[g] = the batch's log-likelihood terms (+0 where there is none)
template <int NB>
__device__ __forceinline__ void split_apply_terms(const SplitBatch<NB>& b, const EkfPose& w, int pl, v2f (&term)[NB])
{
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        term[g] = bc2(0.0f);
        if (!b.any_obs[g]) {   // nothing observed among these 128 landmarks: the means are copied
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                row_store(w.rout, b.off[g][t], 0, b.mx[g][t]);
                row_store(w.rout, b.off[g][t], pl, b.my[g][t]);
            }
        } else if (b.special[g]) {
            split_apply_one<NB, true>(b, g, w, pl, term[g]);
        } else {
            split_apply_one<NB, false>(b, g, w, pl, term[g]);
        }
    }
}

template <int NB>
__device__ __forceinline__ void split_apply(const SplitBatch<NB>& b, const EkfPose& w, int pl, v2f& acc)
{
    v2f term[NB];
    split_apply_terms<NB>(b, w, pl, term);
#pragma unroll
    for (int g = 0; g < NB; ++g) acc = acc + term[g];   // (a batch without observations adds +0: the bits stay)
}

template <int NB, int G, bool OWN_MOTION>
__device__ __forceinline__ void ekf_split_body(const EkfArgs& a, int bid, float (*s_acc)[G][128], const MotionIO& mio,
                                               const MotionParams& mpar)
{
    const unsigned lane = threadIdx.x & 63u;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g0 = (bid * kEkfWaves + wave) * G;
    if (g0 >= a.n) return;
    const int nslots = a.n - g0 < G ? a.n - g0 : G;
    // lane k prepares particle g0 + k: source row, class, pose; read back with v_readlane below
    const int mine = g0 + ((int)lane < nslots ? (int)lane : 0);
    const int src_l = a.anc ? a.anc[mine] : mine;
    if (a.group_filter) {   // sharded: this launch takes the groups fed from local rows only (1) or the others (2)
        const bool remote = __ballot(src_l >= a.n) != 0;
        if (remote != (a.group_filter == 2)) return;
    }
    const int cls_l = a.cls_in[src_l];
    float st_l, ct_l, px_l, py_l;
    if constexpr (OWN_MOTION) {
        // the ancestor's POSE comes through the scorer's index (a sharded session reads it out of the all-gathered poses of
        // every rank; on one GPU the two indices are the same array)
        const int psrc = mio.anc ? mio.anc[mine] : mine;
        float th_l;
        motion_sample_one(mpar, (uint64_t)mine, mio.sx[psrc], mio.sy[psrc], mio.sth[psrc], px_l, py_l, th_l);
        det_sincosf(th_l, st_l, ct_l);
    } else {
        det_sincosf(a.th[mine], st_l, ct_l);
        px_l = a.x[mine];
        py_l = a.y[mine];
    }
    if ((int)lane < nslots) {   // the class follows the particle and is still in use
        a.cls_out[mine] = cls_l;
        a.cstamp[cls_l] = a.stamp_now;
    }
#pragma unroll
    for (int k = 0; k < G; ++k) {
        s_acc[wave][k][lane] = 0.0f;
        s_acc[wave][k][lane + 64] = 0.0f;
    }
    const int pl = __builtin_amdgcn_readfirstlane(a.plane_stride * 4);
    const int mean_bytes = 2 * pl, cov_bytes = 3 * pl;
    const gchar* ozx = uniform_gptr(a.obs_zx);
    const gchar* ozy = uniform_gptr(a.obs_zy);
    const unsigned L = (unsigned)a.nlandmarks, room = (unsigned)a.plane_stride;
    const v2f q2 = bc2(a.meas_var);
    const float nan = __uint_as_float(0x7fc00000u);

    auto pose_of = [&](int k) {
        EkfPose w;
        const int i = g0 + k;
        w.rout = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_gptr(a.map_out + (int64_t)i * a.row_stride), 0, mean_bytes, 0x00020000);
        w.s = bc2(lane_value(st_l, k));
        w.c = bc2(lane_value(ct_l, k));
        w.px = bc2(lane_value(px_l, k));
        w.py = bc2(lane_value(py_l, k));
        return w;
    };

    // What a pass needs from memory before it can start: the observations of its landmarks, the means of the group's first
    // ancestor and the covariance row of its class, all issued together.  (Issuing the loads of pass p + 1 before pass p is
    // worked on was built and measured: 99.2 against 97.7 us for the fused front at 64k x 500, at 44 more VGPRs — the kernel is
    // bound by its vector instructions, 61 us of them at 64k x 500, and by the drain of its row stores, which a load phase
    // behind them has to wait for on this hardware; removed.)
    struct Raw {
        v2f zx[NB], zy[NB], mx[NB], my[NB], pr[NB][5];
        unsigned off[NB][2];
        bool in[NB][2];
    };
    const int src0 = __builtin_amdgcn_readlane(src_l, 0), cls0 = __builtin_amdgcn_readlane(cls_l, 0);
    auto load_means = [&](int src, const unsigned (&off)[NB][2], v2f (&mx)[NB], v2f (&my)[NB]) {
        const __amdgpu_buffer_rsrc_t rin =
            __builtin_amdgcn_make_buffer_rsrc((void*)uniform_gptr(a.map_in + (int64_t)src * a.row_stride), 0, mean_bytes, 0x00020000);
#pragma unroll
        for (int g = 0; g < NB; ++g)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                mx[g][t] = row_load(rin, off[g][t], 0);
                my[g][t] = row_load(rin, off[g][t], pl);
            }
    };
    auto load_cov = [&](int cls, const unsigned (&off)[NB][2], v2f (&pr)[NB][5]) {
        const __amdgpu_buffer_rsrc_t rc =
            __builtin_amdgcn_make_buffer_rsrc((void*)uniform_gptr(a.cov + (int64_t)cls * a.cov_stride), 0, cov_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx =
            __builtin_amdgcn_make_buffer_rsrc((void*)uniform_gptr(a.covx + (int64_t)cls * a.covx_stride), 0, 2 * pl, 0x00020000);
#pragma unroll
        for (int g = 0; g < NB; ++g)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int p = 0; p < 3; ++p) pr[g][p][t] = row_load(rc, off[g][t], p * pl);
                pr[g][3][t] = row_load(rx, off[g][t], 0);
                pr[g][4][t] = row_load(rx, off[g][t], pl);
            }
    };
    auto issue = [&](unsigned lb, Raw& r) {
#pragma unroll
        for (int g = 0; g < NB; ++g)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const unsigned l = lb + (unsigned)g * 128u + 64u * t + lane;
                r.in[g][t] = l < L;
                r.off[g][t] = l < room ? l * 4u : 0xffffffffu;   // beyond the planes: loads give 0, stores are dropped
                const unsigned zo = (r.in[g][t] ? l : 0u) * 4u;
                r.zx[g][t] = *(const gfloat*)(ozx + zo);
                r.zy[g][t] = *(const gfloat*)(ozy + zo);
            }
        load_means(src0, r.off, r.mx, r.my);
        load_cov(cls0, r.off, r.pr);
    };
    // everything about the update that depends on the class's covariances alone
    auto prepare = [&](SplitBatch<NB>& b, const v2f (&pr)[NB][5]) {
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            b.first[g][0] = pr[g][0][0] < 0.0f;
            b.first[g][1] = pr[g][0][1] < 0.0f;
            b.special[g] = b.any_keep[g] || __ballot(b.first[g][0] || b.first[g][1]) != 0;
            if (b.any_obs[g]) b.sh[g] = ekf_shared_from<v2f, false>(pr[g][0], pr[g][1], pr[g][2], q2, pr[g][3], pr[g][4]);
        }
    };

    constexpr unsigned kStep = 128u * NB;
    for (unsigned lb = 0; lb < L; lb += kStep) {
        Raw cur;
        issue(lb, cur);
        SplitBatch<NB> b;
#pragma unroll
        for (int g = 0; g < NB; ++g)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                b.off[g][t] = cur.off[g][t];
                b.zx[g][t] = cur.in[g][t] ? cur.zx[g][t] : nan;
                b.zy[g][t] = cur.in[g][t] ? cur.zy[g][t] : nan;
                b.keep[g][t] = !(b.zx[g][t] == b.zx[g][t] && b.zy[g][t] == b.zy[g][t]);
                b.mx[g][t] = cur.mx[g][t];
                b.my[g][t] = cur.my[g][t];
            }
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            b.any_obs[g] = __ballot(!(b.keep[g][0] && b.keep[g][1])) != 0;
            b.any_keep[g] = __ballot(b.keep[g][0] || b.keep[g][1]) != 0;
        }
        prepare(b, cur.pr);
        int prev = src0, prev_cls = cls0;
        for (int k = 0; k < nslots; ++k) {
            const int src = __builtin_amdgcn_readlane(src_l, k);
            const int cls = __builtin_amdgcn_readlane(cls_l, k);
            if (src != prev) {   // another ancestor: its means into registers (wave-uniform branch)
                load_means(src, b.off, b.mx, b.my);
                prev = src;
            }
            if (cls != prev_cls) {   // another class: its covariances with their determinant terms
                v2f pr[NB][5];
                load_cov(cls, b.off, pr);
                prepare(b, pr);
                prev_cls = cls;
            }
            const EkfPose w = pose_of(k);
            v2f acc = (v2f){s_acc[wave][k][lane], s_acc[wave][k][lane + 64]};
            split_apply<NB>(b, w, pl, acc);
            s_acc[wave][k][lane] = acc[0];
            s_acc[wave][k][lane + 64] = acc[1];
        }
    }
    // the G sums side by side (wave_xor_tree_sum for every particle, the steps interleaved: one after the other they were 6 G
    // dependent cross-lane round trips at the end of every wavefront's life); slots beyond nslots hold zeros
    float tot[G];
#pragma unroll
    for (int k = 0; k < G; ++k) tot[k] = s_acc[wave][k][lane] + s_acc[wave][k][lane + 64];
#pragma unroll
    for (int s = 1; s < 64; s <<= 1)
#pragma unroll
        for (int k = 0; k < G; ++k) tot[k] = tot[k] + __shfl_xor(tot[k], s, 64);
    float total = 0.0f;   // lane k: the sum of particle g0 + k (every lane holds all of them)
#pragma unroll
    for (int k = 0; k < G; ++k) total = (int)lane == k ? tot[k] : total;
    if ((int)lane < nslots) {
        a.loglik[g0 + (int)lane] = total;
        if (a.loglik_user) a.loglik_user[g0 + (int)lane] = total;
    }
}

// (A second form of this update — ONE PASS PER WAVEFRONT: the four wavefronts of a workgroup take the passes of a row side by
// side and share the group's particles, so that no wavefront loads after it has stored; the batches' log-likelihood terms parked
// in LDS and added up in landmark order behind a workgroup barrier, the group's motion samples worked out by one wavefront —
// was built, bit-exact on the whole split suite, and measured slower: fused front 95.5 against 88.6 us at 64k x 500, 2.21
// against 1.80 ms at 1M x 1000 (twice / four times the wavefronts, three barriers per workgroup, 32 KB of LDS that cap the
// occupancy at 4).  Removed; profiles/r04_split_tuning.md.)

template <int NB, int G>
__global__ __launch_bounds__(kEkfWaves * 64) EKF_SPLIT_ATTR void ekf_split_kernel(EkfArgs a)
{
    __shared__ float s_acc[kEkfWaves][G][128];
    int bid = blockIdx.x;
    if (a.xcd_chunk > 0) bid = (bid & 7) * a.xcd_chunk + (bid >> 3);
    ekf_split_body<NB, G, false>(a, bid, s_acc, MotionIO{}, MotionParams{});
}

// ---- the FRONT of a single-GPU frame in one launch: motion sample + scan-match score (score_body.h) and the grouped
// out-of-place landmark update side by side.  The two are bound by different units — the scorer by the texture addresser
// (gathers out of L2), the update by HBM writes — and neither needs the other's output: both start from the resample
// indices and the previous poses (the update works out its particles' motion samples itself).  As two launches they run one
// after the other (a second stream with an event fork and join costs more than it wins: DESIGN.md section 11.5); here the
// workgroups of both kinds are dealt out interleaved — of every `score_octets + ekf_octets` consecutive octets of workgroups
// (an octet = one workgroup per XCD) the scoring ones are spread evenly — so the gathers run in the shadow of the row
// stores.  Same bits as the two launches (same device functions).
struct FrontArgs {
    ScoreGrid g;
    const float *bx, *by;
    int nbeams;
    float* score;
    int32_t* count;
    MotionIO mio;
    MotionParams mpar;
    EkfArgs a;
    int score_blocks;    // 256-thread slices of poses to score
    int score_octets;    // ceil(score_blocks / 8)
    int ekf_octets;      // update workgroups per XCD (the xcd_chunk of ekf_update_group_kernel)
    int score_span;      // the scoring octets lie among the first score_span octets of the grid
};

template <int NB, int G, int LPP, int DEPTH, bool SPLIT = false, bool PACKED = false>
__global__ __launch_bounds__(kEkfWaves * 64) EKF_FRONT_ATTR(SPLIT) void frame_front_kernel(FrontArgs f)
{
    static_assert(kScoreBlock == kEkfWaves * 64, "both kinds of workgroup have 256 threads");
    extern __shared__ float4 s_pair[];
    __shared__ float s_acc[kEkfWaves][G][128];
    const int o = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
    // the scoring octets are spread evenly over the first `span` octets of the grid: the whole grid, unless SLAM_FRONT_SPAN (per
    // cent, measurements) says otherwise — 64k x 500, 4 / 2 particles per updating wavefront: 100 % 130.7 / 158.5 us, 75 %
    // 134.0 / 156.3, 50 % 155.1 / 154.8, 25 % 141.2 / 157.9
    const int64_t span = f.score_span;
    const int before = o < span ? (int)((int64_t)o * f.score_octets / span) : f.score_octets;             // scoring octets among 0 .. o - 1
    const int upto = o + 1 < span ? (int)((int64_t)(o + 1) * f.score_octets / span) : f.score_octets;    // ... among 0 .. o
    if (upto > before) {   // a scoring octet (wave-uniform, workgroup-uniform)
        const int sb = before * 8 + xcd;
        if (sb >= f.score_blocks) return;
        score_poses_body<false, LPP, DEPTH, true, PACKED>(f.g, f.bx, f.by, f.nbeams, f.mio.x, f.mio.y, f.mio.th, nullptr, f.a.n,
                                                          f.score, f.count, f.mio, f.mpar, sb, s_pair);
    } else if constexpr (SPLIT) {
        ekf_split_body<NB, G, true>(f.a, xcd * f.ekf_octets + (o - before), s_acc, f.mio, f.mpar);
    } else {
        ekf_group_body<NB, G, true>(f.a, xcd * f.ekf_octets + (o - before), s_acc, f.mio, f.mpar);
    }
}

// ---- sparse in-place form: frames that keep their population update only the OBSERVED landmarks, in place.
// Walking the rows in batches of 128 (ekf_batches, in place) runs the whole update arithmetic at full wavefront cost for
// the handful of lanes of a batch that hold an observation and touches every line a batch's observed landmarks lie in
// once per batch.  Here the observations are first compacted into a list sorted by landmark (once per observation
// table, build_obs_list_kernel); a wavefront then owns one particle and a LANE owns an observation (two per lane, as
// float2): gather the five values at that landmark, update, scatter them back — the same arithmetic in the same order.
// The log-likelihood keeps the summation order of the specification (landmark l adds to accumulator l mod 128 in order
// of l): the accumulators live in LDS, and observations that fall into the same accumulator carry a round number
// (how many earlier observations share it) and are added round by round.
struct ObsList {
    const int32_t* id;      // [nobs] landmark of observation k, ascending
    const float *zx, *zy;   // [nobs]
    const int32_t* round;   // [nobs] number of earlier observations with the same id mod 128
    const int32_t* count;   // [2] device: nobs, highest round
};

// one workgroup: table (NaN = not observed) -> list in landmark order, rounds, counts (also to mapped host memory).
// L <= kObsListMaxLandmarks (the bitmap of observed landmarks lives in LDS).
__global__ __launch_bounds__(1024) void build_obs_list_kernel(const float* __restrict__ tzx, const float* __restrict__ tzy,
                                                              int L, int32_t* __restrict__ id, float* __restrict__ zx,
                                                              float* __restrict__ zy, int32_t* __restrict__ round,
                                                              int32_t* __restrict__ count, int32_t* __restrict__ h_count)
{
    __shared__ unsigned s_bits[kObsListMaxLandmarks / 32];
    __shared__ int s_wave[16];
    __shared__ int s_base;
    __shared__ int s_max_round;
    if (threadIdx.x == 0) { s_base = 0; s_max_round = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int l0 = 0; l0 < L; l0 += 1024) {   // ordered compaction, 1024 landmarks per step
        const int l = l0 + (int)threadIdx.x;
        const float vx = l < L ? tzx[l] : __builtin_nanf(""), vy = l < L ? tzy[l] : __builtin_nanf("");
        const bool ob = vx == vx && vy == vy;
        const unsigned long long m = __ballot(ob);
        if (lane == 0) s_wave[wave] = __popcll(m);
        if (lane == 0) s_bits[(l0 >> 5) + 2 * wave] = (unsigned)m;
        if (lane == 32) s_bits[(l0 >> 5) + 2 * wave + 1] = (unsigned)(m >> 32);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (ob) {
            const int k = off + __popcll(m & ((1ull << lane) - 1ull));
            id[k] = l;
            zx[k] = vx;
            zy[k] = vy;
            // round: earlier observed landmarks with the same l mod 128 = the same bit of every fourth word below
            int r = 0;
            for (int b = l - 128; b >= 0; b -= 128) r += (int)((s_bits[b >> 5] >> (b & 31)) & 1u);
            round[k] = r;
            if (r > 0) atomicMax(&s_max_round, r);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += s_wave[w];
            s_base += tot;
        }
        __syncthreads();
    }
    const int nobs = s_base;
    __syncthreads();
    if (threadIdx.x == 0) {
        count[0] = nobs;
        count[1] = s_max_round;
        if (h_count) {
            h_count[0] = nobs;
            h_count[1] = L;
        }
    }
}

__global__ __launch_bounds__(kEkfWaves * 64) void ekf_sparse_kernel(EkfArgs a, ObsList ol)
{
    __shared__ float s_acc[kEkfWaves][128];
    const unsigned lane = threadIdx.x & 63u;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int bid = blockIdx.x;
    if (a.xcd_chunk > 0) bid = (bid & 7) * a.xcd_chunk + (bid >> 3);
    const int i = bid * kEkfWaves + wave;
    if (i >= a.n) return;
    float st_, ct_;
    det_sincosf(a.th[i], st_, ct_);
    const int row_bytes = __builtin_amdgcn_readfirstlane(5 * a.plane_stride * 4);
    const int pl = __builtin_amdgcn_readfirstlane(a.plane_stride * 4);
    const __amdgpu_buffer_rsrc_t row =   // in place: the particle's own row, read and written
        __builtin_amdgcn_make_buffer_rsrc((void*)uniform_gptr(a.map_out + (int64_t)i * a.row_stride), 0, row_bytes, 0x00020000);
    const v2f s = bc2(st_), c = bc2(ct_), px = bc2(a.x[i]), py = bc2(a.y[i]), q = bc2(a.meas_var);
    const int nobs = __builtin_amdgcn_readfirstlane(ol.count[0]);
    const int max_round = __builtin_amdgcn_readfirstlane(ol.count[1]);
    s_acc[wave][lane] = 0.0f;
    s_acc[wave][lane + 64] = 0.0f;
    for (int k0 = 0; k0 < nobs; k0 += 128) {
        bool ob[2];
        unsigned off[2];
        int slot[2], rnd[2];
        v2f zx, zy, m[5];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int k = k0 + 64 * t + (int)lane;
            ob[t] = k < nobs;
            const int kk = ob[t] ? k : 0;
            const int l = ol.id[kk];
            off[t] = (unsigned)l * 4u;
            slot[t] = l & 127;
            rnd[t] = ol.round[kk];
            zx[t] = ol.zx[kk];
            zy[t] = ol.zy[kk];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 5; ++p) m[p][t] = ob[t] ? row_load(row, off[t], p * pl) : 1.0f;   // 1: harmless operands for idle lanes
        const v2f mx = m[0], my = m[1], pxx = m[2], pxy = m[3], pyy = m[4];
        const EkfResult<v2f> u = ekf_update_one<v2f>(mx, my, pxx, pxy, pyy, zx, zy, s, c, px, py, q);
        const v2f o0 = u.o0, o1 = u.o1, o2 = u.o2, o3 = u.o3, o4 = u.o4, f0 = u.f0, f1 = u.f1, ll = u.ll;
        float term[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bool first = pxx[t] < 0.0f;
            term[t] = first ? 0.0f : ll[t];
            if (ob[t]) {
                row_store(row, off[t], 0 * pl, first ? f0[t] : o0[t]);
                row_store(row, off[t], 1 * pl, first ? f1[t] : o1[t]);
                row_store(row, off[t], 2 * pl, first ? q[t] : o2[t]);
                row_store(row, off[t], 3 * pl, first ? 0.0f : o3[t]);
                row_store(row, off[t], 4 * pl, first ? q[t] : o4[t]);
            }
        }
        // log-likelihood: accumulator = landmark mod 128, in order of the landmark: round by round (observations of one
        // accumulator have distinct rounds; a wavefront's LDS operations execute in order)
        for (int r = 0; r <= max_round; ++r)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (ob[t] && rnd[t] == r) s_acc[wave][slot[t]] = s_acc[wave][slot[t]] + term[t];
    }
    const float total = wave_xor_tree_sum(s_acc[wave][lane] + s_acc[wave][lane + 64]);
    if (lane == 0) {
        a.loglik[i] = total;
        if (a.loglik_user) a.loglik_user[i] = total;
    }
}

// ------------------------------------------------------------------ A11: weights
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// carry (optional): the normalised log-weights the previous frame left behind; they count only when that frame did
// NOT resample (*prev_resampled == 0, a device flag written by the previous frame's resample kernel)
__global__ __launch_bounds__(kBlock) void logweight_kernel(const float* __restrict__ score,
                                                           const float* __restrict__ loglik, float gain, int n,
                                                           const float* __restrict__ carry,
                                                           const int32_t* __restrict__ prev_resampled,
                                                           float* __restrict__ logw, float* __restrict__ block_max)
{
    __shared__ float s_max[kBlock / 64];
    float m = -INFINITY;
    const bool add_carry = carry && *prev_resampled == 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float ll = loglik ? loglik[i] : 0.0f;
        const float sc = score ? score[i] * gain : 0.0f;
        float lw = ll - sc;
        if (add_carry) lw = carry[i] + lw;
        logw[i] = lw;
        m = lw > m ? lw : m;
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) m = fmaxf(m, s_max[w]);
        block_max[blockIdx.x] = m;
    }
}

// The same launch with the covariance classes' update of a split session in workgroups of its own behind the first `nw`
// (cov_update_body.h): the two have nothing to do with each other — which is the point, they need no launch each.
__global__ __launch_bounds__(kBlock) void logweight_cov_kernel(const float* __restrict__ score, const float* __restrict__ loglik,
                                                               float gain, int n, const float* __restrict__ carry,
                                                               const int32_t* __restrict__ prev_resampled, float* __restrict__ logw,
                                                               float* __restrict__ block_max, int nw, int ly, CovArgs cov)
{
    if ((int)blockIdx.x >= nw) {
        const int j = (int)blockIdx.x - nw;
        cov_update_body(cov, j / ly, j % ly);
        return;
    }
    __shared__ float s_max[kBlock / 64];
    float m = -INFINITY;
    const bool add_carry = carry && *prev_resampled == 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += nw * kBlock) {
        const float ll = loglik ? loglik[i] : 0.0f;
        const float sc = score ? score[i] * gain : 0.0f;
        float lw = ll - sc;
        if (add_carry) lw = carry[i] + lw;
        logw[i] = lw;
        m = lw > m ? lw : m;
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) m = fmaxf(m, s_max[w]);
        block_max[blockIdx.x] = m;
    }
}

// Several GPUs: the maximum of the block maxima as one float for the all-reduce.  (Folding this into the kernel above with
// a "last workgroup done" ticket was measured and dropped: two atomics per workgroup on one address run at ~88 per
// microsecond, 47 us at 2048 workgroups against 5 us for this launch.)
__global__ __launch_bounds__(kBlock) void max_finalize_kernel(const float* __restrict__ block_max, int nblocks,
                                                              float* __restrict__ d_max)
{
    __shared__ float s_max[kBlock / 64];
    float m = -INFINITY;
    for (int i = threadIdx.x; i < nblocks; i += kBlock) m = fmaxf(m, block_max[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) m = fmaxf(m, s_max[w]);
        *d_max = m;
    }
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, o), hi = __shfl_xor((uint32_t)(v >> 32), o);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

__global__ __launch_bounds__(kBlock) void quantise_weights_kernel(const float* __restrict__ logw,
                                                                  const float* __restrict__ d_max, int n,
                                                                  uint64_t* __restrict__ wq,
                                                                  unsigned long long* __restrict__ d_sum)
{
    __shared__ uint64_t s_sum[kBlock / 64];
    const float m = *d_max;
    uint64_t acc = 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float w = det_expf(logw[i] - m);
        const uint64_t q = (uint64_t)(w * 4294967296.0f);
        wq[i] = q;
        acc += q;
    }
    acc = wave_sum_u64(acc);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) acc += s_sum[w];
        atomicAdd(d_sum, (unsigned long long)acc);   // integer: exact and order-independent
    }
}

// ------------------------------------------------------------------ A12: integer CDF, comb, ancestors
constexpr int kScanItems = 8;                       // elements per thread
constexpr int kScanTile = kBlock * kScanItems;      // 2048 elements per workgroup

__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int d)
{
    const uint32_t lo = __shfl_up((uint32_t)v, d), hi = __shfl_up((uint32_t)(v >> 32), d);
    return ((uint64_t)hi << 32) | lo;
}

// inclusive scan of one value per thread over the workgroup: wavefront scan + carry through LDS
__device__ __forceinline__ uint64_t block_inclusive_scan(uint64_t v, uint64_t* s_wave /*[kBlock/64]*/, uint64_t& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t up = shfl_up_u64(v, d);
        if (lane >= d) v += up;
    }
    if (lane == 63) s_wave[wave] = v;
    __syncthreads();
    uint64_t carry = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
        if (w < wave) carry += s_wave[w];
        tot += s_wave[w];
    }
    total = tot;
    __syncthreads();
    return v + carry;
}

__global__ __launch_bounds__(kBlock) void scan_tiles_kernel(const uint64_t* __restrict__ in, int n,
                                                            uint64_t* __restrict__ out,
                                                            uint64_t* __restrict__ tile_total)
{
    __shared__ uint64_t s_wave[kBlock / 64];
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint64_t v[kScanItems];
    uint64_t run = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int i = base + k;
        run += i < n ? in[i] : 0;
        v[k] = run;
    }
    uint64_t total;
    const uint64_t incl = block_inclusive_scan(run, s_wave, total);
    const uint64_t excl = incl - run;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int i = base + k;
        if (i < n) out[i] = v[k] + excl;
    }
    if (threadIdx.x == 0) tile_total[blockIdx.x] = total;
}

// exclusive scan of the tile totals, in place, by ONE workgroup (<= a few thousand tiles)
__global__ __launch_bounds__(kBlock) void scan_totals_kernel(uint64_t* __restrict__ tile_total, int ntiles)
{
    __shared__ uint64_t s_wave[kBlock / 64];
    uint64_t carry = 0;
    for (int t0 = 0; t0 < ntiles; t0 += kBlock) {
        const int t = t0 + threadIdx.x;
        const uint64_t v = t < ntiles ? tile_total[t] : 0;
        uint64_t total;
        const uint64_t incl = block_inclusive_scan(v, s_wave, total);
        if (t < ntiles) tile_total[t] = carry + incl - v;
        carry += total;
    }
}

__global__ __launch_bounds__(kBlock) void add_tile_offsets_kernel(uint64_t* __restrict__ out, int n,
                                                                  const uint64_t* __restrict__ tile_excl)
{
    const uint64_t off = tile_excl[blockIdx.x];
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int i = base + k;
        if (i < n) out[i] += off;
    }
}

// ---- fused frame-loop form: weights are quantised and scanned in one pass, never stored
// max over an array of block maxima, by the whole workgroup (every workgroup repeats it: <= 2048 floats)
__device__ __forceinline__ float block_max_of(const float* __restrict__ v, int count, float* s_red /*[kBlock/64]*/)
{
    float m = -INFINITY;
    for (int i = threadIdx.x; i < count; i += kBlock) m = fmaxf(m, v[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    float r = s_red[0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) r = fmaxf(r, s_red[w]);
    __syncthreads();
    return r;
}

__device__ __forceinline__ uint64_t block_sum_u64(uint64_t v, uint64_t* s_red /*[kBlock/64]*/)
{
    v = wave_sum_u64(v);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    uint64_t r = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) r += s_red[w];
    __syncthreads();
    return r;
}

// GATED: also what the resample gate needs — the 16-bit weight sums S = sum(wq >> 16), Q = sum((wq >> 16)^2) of the
// tile (exact integers, hence independent of order and sharding) and carry[i] = logw[i] - max, the weight a particle
// takes into the next frame when this one does not resample (oracle: orc_ess_terms / orc_weight_carry).
template <bool GATED>
__global__ __launch_bounds__(kBlock) void quantise_scan_kernel(const float* __restrict__ logw,
                                                               const float* __restrict__ d_max,
                                                               const float* __restrict__ block_max, int nblock_max,
                                                               int n, uint64_t* __restrict__ cdf_local,
                                                               uint64_t* __restrict__ tile_total,
                                                               float* __restrict__ carry,
                                                               uint64_t* __restrict__ tile_s16,
                                                               uint64_t* __restrict__ tile_q16,
                                                               uint64_t* __restrict__ d_sum,
                                                               unsigned int* __restrict__ ticket)
{
    __shared__ uint64_t s_wave[kBlock / 64];
    __shared__ float s_red[kBlock / 64];
    const float m = d_max ? *d_max : block_max_of(block_max, nblock_max, s_red);
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint64_t v[kScanItems];
    uint64_t run = 0, s16 = 0, q16 = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int i = base + k;
        uint64_t q = 0;
        if (i < n) {
            const float rel = logw[i] - m;
            q = (uint64_t)(det_expf(rel) * 4294967296.0f);
            if (GATED) carry[i] = rel;
        }
        run += q;
        v[k] = run;
        if (GATED) {
            const uint64_t w = q >> 16;
            s16 += w;
            q16 += w * w;
        }
    }
    uint64_t total;
    const uint64_t incl = block_inclusive_scan(run, s_wave, total);
    const uint64_t excl = incl - run;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int i = base + k;
        if (i < n) cdf_local[i] = v[k] + excl;   // inclusive, local to this 2048-element tile
    }
    if (threadIdx.x == 0) tile_total[blockIdx.x] = total;
    if (GATED) {
        s16 = wave_sum_u64(s16);
        q16 = wave_sum_u64(q16);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = s16;
        __syncthreads();
        uint64_t a = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) a += s_wave[w];
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = q16;
        __syncthreads();
        uint64_t b = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) b += s_wave[w];
        if (threadIdx.x == 0) {
            tile_s16[blockIdx.x] = a;
            tile_q16[blockIdx.x] = b;
        }
    }
    if (!d_sum || threadIdx.x != 0) return;
    // several GPUs: the shard's sums (what the ranks all-gather) through integer atomics; the ticket add depends on their
    // return values, so it is issued only after they have landed; the workgroup that finishes last hands the sums over and
    // clears the accumulators.  No fences: nothing but atomics is published (a release fence per workgroup — an L2
    // write-back across the XCDs — made this kernel 33 instead of 12 us at 512 workgroups).  Saves the one-workgroup launch
    // behind this kernel at the usual shard sizes (32 workgroups at 64k particles).
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(ticket + 2);   // 3 accumulators behind the ticket
    unsigned long long dep = atomicAdd(&acc[0], (unsigned long long)total);
    if (GATED) {
        dep ^= atomicAdd(&acc[1], (unsigned long long)tile_s16[blockIdx.x]);
        dep ^= atomicAdd(&acc[2], (unsigned long long)tile_q16[blockIdx.x]);
    }
    uint32_t one = 1u;
    asm volatile("" : "+v"(one) : "v"(dep));
    if (atomicAdd(&ticket[0], one) != gridDim.x - 1) return;
    for (int a = 0; a < (GATED ? 3 : 1); ++a) d_sum[a] = atomicExch(&acc[a], 0ull);
    ticket[0] = 0;
}

// The resample gate (oracle: orc_ess_resample): resample iff ESS < frac * N, i.e. S^2 * 65536 < frac_q16 * N * Q, in
// 128-bit integer arithmetic (S < 2^47, Q < 2^63, N < 2^31, frac_q16 <= 2^16).
__device__ __forceinline__ bool ess_wants_resample(uint64_t s16, uint64_t q16, uint64_t n_total, uint32_t frac_q16)
{
    uint64_t l_lo = s16 * s16, l_hi = __umul64hi(s16, s16);
    l_hi = (l_hi << 16) | (l_lo >> 48);
    l_lo <<= 16;
    const uint64_t nf = n_total * (uint64_t)frac_q16;
    const uint64_t r_lo = q16 * nf, r_hi = __umul64hi(q16, nf);
    return l_hi < r_hi || (l_hi == r_hi && l_lo < r_lo);
}

// verdict of the gate for the rest of the frame loop: a device flag (the next frame's weight kernel reads it) and the
// same value in mapped host memory behind a sequence number (the host picks the EKF form for the next frame)
__device__ __forceinline__ void publish_gate(const GateOut& g, bool resample)
{
    *g.d_flag = resample ? 1 : 0;
    if (g.h_flag) {
        g.h_flag[0] = resample ? 1 : 0;
        __threadfence_system();
        __hip_atomic_store(reinterpret_cast<uint32_t*>(g.h_flag + 1), g.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// floor((hi:lo) / d) for hi < d < 2^63 (so the quotient fits 64 bits): restoring long division
__device__ __forceinline__ uint64_t div128by64(uint64_t hi, uint64_t lo, uint64_t d)
{
    uint64_t rem = hi, q = 0;
#pragma unroll 8
    for (int b = 63; b >= 0; --b) {
        rem = (rem << 1) | ((lo >> b) & 1ull);
        if (rem >= d) {
            rem -= d;
            q |= 1ull << b;
        }
    }
    return q;
}

__global__ __launch_bounds__(kBlock) void offspring_offsets_kernel(const uint64_t* __restrict__ cdf, int n,
                                                                   const uint64_t* __restrict__ d_base,
                                                                   const uint64_t* __restrict__ d_total,
                                                                   uint32_t key0, uint32_t key1, uint32_t frame,
                                                                   uint64_t n_total, int32_t* __restrict__ first)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint64_t base = d_base ? *d_base : 0ull;
    const uint64_t total = *d_total;
    if (total == 0 || (total >> 63)) {   // impossible with finite log-weights (the best particle has wq = 2^32);
        first[i] = 0;                    // stay memory-safe anyway: no division, every slot gets the last particle
        return;
    }
    // comb offset u in [0,total): Philox counter (0,0,frame,1), high 64 bits of r64*total (wave-uniform)
    const u32x4 r = philox4x32_10(0u, 0u, frame, 1u /* resample stream */, key0, key1);
    const uint64_t comb_u = __umul64hi((uint64_t)r.v[0] | ((uint64_t)r.v[1] << 32), total);
    const uint64_t c_excl = base + (i ? cdf[i - 1] : 0ull);
    // X = c_excl * n_total as 128 bits
    uint64_t lo = c_excl * n_total, hi = __umul64hi(c_excl, n_total);
    int32_t f = 0;
    if (hi != 0 || lo > comb_u) {
        // (X - u - 1) / total + 1
        const uint64_t sub = comb_u + 1ull;   // comb_u < total < 2^63: no overflow
        hi -= lo < sub ? 1ull : 0ull;
        lo -= sub;
        f = (int32_t)(div128by64(hi, lo, total) + 1ull);
    }
    first[i] = f;
}

// comb_first(): shared by the staged and the fused offsets kernels
__device__ __forceinline__ int32_t comb_first(uint64_t c_excl, uint64_t total, uint64_t n_total, uint32_t key0,
                                              uint32_t key1, uint32_t frame)
{
    if (total == 0 || (total >> 63)) return 0;   // see offspring_offsets_kernel
    const u32x4 r = philox4x32_10(0u, 0u, frame, 1u /* resample stream */, key0, key1);
    const uint64_t comb_u = __umul64hi((uint64_t)r.v[0] | ((uint64_t)r.v[1] << 32), total);
    uint64_t lo = c_excl * n_total, hi = __umul64hi(c_excl, n_total);
    if (hi == 0 && lo <= comb_u) return 0;
    const uint64_t sub = comb_u + 1ull;
    hi -= lo < sub ? 1ull : 0ull;
    lo -= sub;
    return (int32_t)(div128by64(hi, lo, total) + 1ull);
}

// fused form: CDF = base + (sum of earlier tiles) + tile-local scan; every workgroup re-derives its tile's
// offset (and, on a single GPU, the grand total) from the <= n/2048 tile totals instead of a separate pass
__global__ __launch_bounds__(kBlock) void offspring_from_scan_kernel(const uint64_t* __restrict__ cdf_local,
                                                                     const uint64_t* __restrict__ tile_total,
                                                                     int ntiles, int n,
                                                                     const uint64_t* __restrict__ d_base,
                                                                     const uint64_t* __restrict__ d_total,
                                                                     const uint64_t* __restrict__ d_shard_totals,
                                                                     int rank, int world, uint32_t key0,
                                                                     uint32_t key1, uint32_t frame, uint64_t n_total,
                                                                     int32_t* __restrict__ first,
                                                                     const uint64_t* __restrict__ tile_s16,
                                                                     const uint64_t* __restrict__ tile_q16,
                                                                     uint32_t frac_q16, GateOut gate)
{
    __shared__ uint64_t s_red[kBlock / 64];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int tile = (blockIdx.x * kBlock) / kScanTile;   // kScanTile is a multiple of kBlock
    const bool gated = frac_q16 != 0;
    const int stride = gated ? 3 : 1;   // gated: the ranks all-gather (total, S, Q) triples
    uint64_t before = 0, all = 0, s16 = 0, q16 = 0;
    for (int t = threadIdx.x; t < ntiles; t += kBlock) {
        const uint64_t v = tile_total[t];
        all += v;
        before += t < tile ? v : 0ull;
        if (gated && !d_shard_totals) {
            s16 += tile_s16[t];
            q16 += tile_q16[t];
        }
    }
    before = block_sum_u64(before, s_red);
    uint64_t total, shard_base = d_base ? *d_base : 0ull;
    if (d_shard_totals) {   // several GPUs: the all-gathered shard totals give both the base and the grand total
        total = 0;
        shard_base = 0;
        s16 = q16 = 0;
        for (int q = 0; q < world; ++q) {
            const uint64_t v = d_shard_totals[(size_t)stride * q];
            total += v;
            shard_base += q < rank ? v : 0ull;
            if (gated) {
                s16 += d_shard_totals[3 * (size_t)q + 1];
                q16 += d_shard_totals[3 * (size_t)q + 2];
            }
        }
    } else {
        total = d_total ? *d_total : block_sum_u64(all, s_red);
        if (gated) {
            s16 = block_sum_u64(s16, s_red);
            q16 = block_sum_u64(q16, s_red);
        }
    }
    if (gated) {   // the same verdict in every workgroup and on every rank (integer sums over the whole population)
        const bool resample = ess_wants_resample(s16, q16, n_total, frac_q16);
        if (blockIdx.x == 0 && threadIdx.x == 0) publish_gate(gate, resample);
        if (!resample) {   // every particle keeps its slot: slot j descends from particle j
            if (i < n) first[i] = (int32_t)((int64_t)rank * n + i);
            return;
        }
    }
    if (i >= n) return;
    const uint64_t base = shard_base + before;
    const uint64_t c_excl = base + ((i % kScanTile) ? cdf_local[i - 1] : 0ull);
    first[i] = comb_first(c_excl, total, n_total, key0, key1, frame);
}

// Feedback for the host's choice between the two out-of-place EKF forms (it changes speed, never results): about how many
// DISTINCT ancestors the resample left — a slot counts when its wavefront neighbour descends from another particle (so
// wavefront boundaries count once too often: at most n / 64).  Summed with one atomic per workgroup; the workgroup that
// finishes last hands {count, n} to mapped host memory and clears the counter (8-byte aligned pair of words).  The host
// reads it without any synchronisation, a frame or two late.
__device__ __forceinline__ void count_heads(const HeadsOut& h, int val, bool valid, int n)
{
    // a sample is enough for a heuristic: every 8th workgroup counts, the result is scaled (same-address atomics run at
    // ~88 per microsecond: one per workgroup cost 14 us at 4096 workgroups)
    if (!h.counter || (blockIdx.x & 7u) != 0) return;
    __shared__ int s_heads[kBlock / 64];
    const int up = __shfl_up(val, 1, 64);
    const bool head = valid && ((threadIdx.x & 63) == 0 || up != val);
    const int cnt = __popcll(__ballot(head));
    if ((threadIdx.x & 63) == 0) s_heads[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x != 0) return;
    int sum = 0;
    for (int w = 0; w < kBlock / 64; ++w) sum += s_heads[w];
    // ONE 64-bit atomic carries both the count (high word) and the number of workgroups done (low word): nothing else is
    // published, so no fence is needed (a __threadfence() per workgroup made this kernel 2.5x slower at 1M slots)
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(h.counter);
    const unsigned long long old = atomicAdd(acc, ((unsigned long long)(unsigned)sum << 32) | 1ull);
    const unsigned sampled = (gridDim.x + 7u) / 8u;
    if ((unsigned)(old & 0xffffffffu) != sampled - 1) return;
    const long long slots = (long long)sampled * kBlock < n ? (long long)sampled * kBlock : n;   // slots the sample covered (about)
    h.h_out[0] = (int32_t)(((long long)(old >> 32) + sum) * n / slots);
    h.h_out[1] = n;
    atomicExch(acc, 0ull);
}

// Single GPU: the ancestor of every slot straight from the tile-local scan, without materialising `first`.
// first[i] <= j  <=>  N*C_excl(i) <= j*S + u  (first[i] = ceil((N*C_excl(i) - u)/S), clamped at 0), so the ancestor
// of slot j — the last i with first[i] <= j — is the number of k in [0, n-1) with N*C_incl(k) <= j*S + u.
// Both sides are 96-bit quantities, compared as (hi, lo) pairs.  Tile offsets live in LDS: n <= kMaxLdsTiles*2048.
constexpr int kMaxLdsTiles = 4096;
template <int kPer>   // tiles per thread: 1 covers n <= 512k with 2 KB of LDS, 16 covers n <= 8M with 32 KB
__global__ __launch_bounds__(kBlock) void ancestors_from_scan_kernel(const uint64_t* __restrict__ cdf_local,
                                                                     const uint64_t* __restrict__ tile_total,
                                                                     int ntiles, int n, uint32_t key0, uint32_t key1,
                                                                     uint32_t frame, int32_t* __restrict__ anc,
                                                                     const uint64_t* __restrict__ tile_s16,
                                                                     const uint64_t* __restrict__ tile_q16,
                                                                     uint32_t frac_q16, GateOut gate, HeadsOut heads)
{
    __shared__ uint64_t s_off[kPer * kBlock];
    __shared__ uint64_t s_wave[kBlock / 64];
    uint64_t v[kPer], run = 0, s16 = 0, q16 = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int t = threadIdx.x * kPer + k;
        v[k] = run;   // exclusive within the thread
        run += t < ntiles ? tile_total[t] : 0ull;
        if (frac_q16 != 0 && t < ntiles) {
            s16 += tile_s16[t];
            q16 += tile_q16[t];
        }
    }
    uint64_t total;
    const uint64_t excl = block_inclusive_scan(run, s_wave, total) - run;
#pragma unroll
    for (int k = 0; k < kPer; ++k) s_off[threadIdx.x * kPer + k] = v[k] + excl;
    __syncthreads();
    const int j = blockIdx.x * kBlock + threadIdx.x;
    bool resample = true;
    if (frac_q16 != 0) {   // resample gate: the same verdict in every workgroup
        s16 = block_sum_u64(s16, s_wave);
        q16 = block_sum_u64(q16, s_wave);
        resample = ess_wants_resample(s16, q16, (uint64_t)n, frac_q16);
        if (blockIdx.x == 0 && threadIdx.x == 0) publish_gate(gate, resample);
    }
    int val = -1;   // this slot's ancestor (-1: no such slot)
    const bool search = resample && !(total == 0 || (total >> 63));   // the same in every thread of every workgroup
    if (j < n && !search) val = resample ? n - 1   // see offspring_offsets_kernel: every slot gets the last particle
                                         : j;      // the frame keeps its population
    if (search) {
        const bool in = j < n;
        const uint64_t N = (uint64_t)n;
        uint64_t t_lo = 0, t_hi = 0;
        int tlo = 0;
        auto below = [&](uint64_t c) {   // N * c <= T, both sides 96-bit quantities compared as (hi, lo) pairs
            const uint64_t x_lo = c * N, x_hi = __umul64hi(c, N);
            return x_hi < t_hi || (x_hi == t_hi && x_lo <= t_lo);
        };
        if (in) {
            const u32x4 r = philox4x32_10(0u, 0u, frame, 1u /* resample stream */, key0, key1);
            const uint64_t comb_u = __umul64hi((uint64_t)r.v[0] | ((uint64_t)r.v[1] << 32), total);
            t_lo = (uint64_t)j * total;
            t_hi = __umul64hi((uint64_t)j, total);
            t_lo += comb_u;
            t_hi += t_lo < comb_u ? 1ull : 0ull;
            // number of k in [0, n-1) with N*C_incl(k) <= T.  First among the tiles, in LDS: the CDF at the end of tile t is
            // the offset of tile t + 1 (the grand total for the last one), so whole tiles below T are counted without
            // touching memory; then inside the one tile that holds the boundary.
            int thi = ntiles - 1;   // first tile whose last element lies above T (the last tile if none does)
            while (tlo < thi) {
                const int mid = (tlo + thi) >> 1;
                if (below(s_off[mid + 1])) tlo = mid + 1; else thi = mid;   // mid + 1 <= ntiles - 1
            }
        }
        int lo = tlo * kScanTile, hi = (tlo + 1) * kScanTile < n - 1 ? (tlo + 1) * kScanTile : n - 1;
        const uint64_t off = s_off[tlo];   // (every pivot below lies in tile tlo)
        // kPer == 1 (n <= 512k): a second level in LDS.  The thresholds rise with the slot, so the workgroup's 256 slots fall
        // into the tiles of its first and its last slot; when that is at most two tiles, the CDF at the END of each of their
        // 32-element blocks (64 values per tile: one strided load per thread, one round trip) is staged in LDS, six LDS steps
        // find the block, and five dependent L2 round trips remain of the eleven (measured: 7.6 -> 7.1 us at 64k slots; staging the
        // whole window instead, 16-32 KB, was slower: 9.5 us).  The same comparisons on the same values: the block whose last element is the first above T holds the
        // answer, and if none is, the search ends at the tile's upper bound as it does without the second level.
        if constexpr (kPer == 1) {
            constexpr int kSub = 32, kSubPerTile = kScanTile / kSub;
            __shared__ uint64_t s_sub[2 * kSubPerTile];
            __shared__ int s_t[2];
            const int last = (n - 1 - (int)blockIdx.x * kBlock) < kBlock - 1 ? (n - 1 - (int)blockIdx.x * kBlock) : kBlock - 1;
            if (threadIdx.x == 0) s_t[0] = tlo;
            if ((int)threadIdx.x == last) s_t[1] = tlo;
            __syncthreads();
            const int tmin = s_t[0], tmax = s_t[1];
            const bool staged = tmax - tmin <= 1;   // (workgroup-uniform)
            if (staged) {
                for (int q = threadIdx.x; q < (tmax - tmin + 1) * kSubPerTile; q += kBlock) {
                    const int idx = tmin * kScanTile + q * kSub + kSub - 1;
                    s_sub[q] = cdf_local[idx < n ? idx : n - 1];   // (beyond the data: never looked at below)
                }
                __syncthreads();
                if (in) {
                    // first block b of tile tlo whose last element is above T, among the blocks that end below `hi`
                    const int nblk = (hi - lo) / kSub;   // whole blocks in [lo, hi): their last elements are < hi
                    const uint64_t* sub = s_sub + (tlo - tmin) * kSubPerTile;
                    int blo = 0, bhi = nblk;
                    while (blo < bhi) {
                        const int mid = (blo + bhi) >> 1;
                        if (below(sub[mid] + off)) blo = mid + 1; else bhi = mid;
                    }
                    if (blo < nblk) hi = lo + blo * kSub + kSub - 1;   // that element is above T: the answer is at or below it
                    lo += blo * kSub;
                }
            }
        }
        if (in) {
            // (an 8-ary search — seven pivots per step side by side — was measured equal: 10.5 us at 64k slots, 62 us at 1M:
            // fewer dependent round trips, but seven times the gathers)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (below(cdf_local[mid] + off)) lo = mid + 1; else hi = mid;
            }
            val = lo;
        }
    }
    if (j < n) anc[j] = val;
    count_heads(heads, val, j < n, n);
}

__global__ __launch_bounds__(kBlock) void ancestors_kernel(const int32_t* __restrict__ first_all, int64_t n_total,
                                                           int64_t slot0, int nslots, int32_t* __restrict__ anc)
{
    const int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= nslots) return;
    const int64_t j = slot0 + s;
    int64_t lo = 0, hi = n_total;   // first index whose first slot is > j
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)first_all[mid] <= j) lo = mid + 1; else hi = mid;
    }
    anc[s] = (int32_t)(lo - 1);
}

// ------------------------------------------------------------------ multi-GPU resample: sharded gather index + migration
// Slots of rank r are [r*n, (r+1)*n).  The particles of rank s fill the slot range [A_s, B_s) with
// A_s = first_all[s*n], B_s = A_(s+1) (B of the last rank = n_total) because `first` is non-decreasing.
// So what rank r receives from rank s is ONE contiguous run of its slots, and everything below follows
// from the world+1 boundary values — no host-computed plan is needed for the index kernel.
__device__ __forceinline__ int64_t first_or_total(const int32_t* __restrict__ first_all, int64_t idx, int64_t n_total)
{
    return idx < n_total ? (int64_t)first_all[idx] : n_total;
}

__device__ __forceinline__ int64_t last_with_first_le(const int32_t* __restrict__ first_all, int64_t n_total, int64_t j)
{
    int64_t lo = 0, hi = n_total;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)first_all[mid] <= j) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}

// ---- exchange with duplicates removed.  After a resample many slots share one ancestor; a remote ancestor's row
// is sent ONCE per destination rank, however many of that rank's slots descend from it.  Both sides derive the
// same order from first_all alone:
//   receiver r: D(j) = number of "heads" among its slots up to j (a slot is a head when it is the first of r's
//               slots with that ancestor); the rows received from rank s land in the staging tail in ancestor
//               order, and slot j finds its row at  n + roff[s] + D(j) - D(first slot served by s);
//   sender  s:  P(a) = number of its particles up to a that have any offspring; the rows for rank d are the
//               particles with offspring in d's slot range, in order: the q-th is the first a with
//               P(a) = P(first ancestor of the run) + q.
// Both counts are prefix sums over n elements (two-level: 2048-element tiles, then the tile totals).
constexpr int kShardTile = 2048;                       // elements per workgroup in the flag scan
constexpr int kShardItems = kShardTile / kBlock;       // per thread

// global ancestor of each of my slots: one thread per slot
__global__ __launch_bounds__(kBlock) void shard_search_kernel(const int32_t* __restrict__ first_all, int64_t n_total, int n,
                                                              int rank, int32_t* __restrict__ gsrc)
{
    const int jl = blockIdx.x * kBlock + threadIdx.x;
    if (jl < n) gsrc[jl] = (int32_t)last_with_first_le(first_all, n_total, (int64_t)rank * n + jl);
}

// y = 0: head flags of my slots;  y = 1: "has offspring" flags of my particles
__global__ __launch_bounds__(kBlock) void shard_flag_scan_kernel(const int32_t* __restrict__ first_all, int64_t n_total,
                                                                 int n, int rank, const int32_t* __restrict__ gsrc,
                                                                 int32_t* __restrict__ pfx, int32_t* __restrict__ btot,
                                                                 int ntiles)
{
    __shared__ int32_t s_wave[kBlock / 64];
    const int y = blockIdx.y;
    const int64_t my_lo = (int64_t)rank * n;
    const int base = blockIdx.x * kShardTile + threadIdx.x * kShardItems;
    int32_t f[kShardItems];
    int32_t run = 0;
    if (y == 0) {
        // the ancestors were found by shard_search_kernel, one thread per slot (a thread doing its kShardItems
        // searches itself is 136 dependent L2 round trips: 20 us for this kernel instead of 5 + 5)
#pragma unroll
        for (int k = 0; k < kShardItems; ++k) {
            const int idx = base + k;
            int32_t flag = 0;
            if (idx < n) {
                const int64_t g = gsrc[idx];
                flag = (idx == 0 || (int64_t)first_all[g] == my_lo + idx) ? 1 : 0;
            }
            run += flag;
            f[k] = run;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kShardItems; ++k) {
            const int idx = base + k;
            int32_t flag = 0;
            if (idx < n) {
                const int64_t j = my_lo + idx;
                flag = first_or_total(first_all, j + 1, n_total) > (int64_t)first_all[j] ? 1 : 0;
            }
            run += flag;
            f[k] = run;   // inclusive within the thread
        }
    }
    // exclusive offset of this thread inside the tile: wave scan + wave totals through LDS
    int32_t incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t v = __shfl_up(incl, o, 64);
        if ((int)(threadIdx.x & 63) >= o) incl += v;
    }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    int32_t woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += s_wave[w];
    const int32_t excl = woff + incl - run;
#pragma unroll
    for (int k = 0; k < kShardItems; ++k)
        if (base + k < n) pfx[(int64_t)y * n + base + k] = excl + f[k];
    if (threadIdx.x == kBlock - 1) btot[y * ntiles + blockIdx.x] = woff + incl;
}

__device__ __forceinline__ int32_t shard_prefix(const int32_t* __restrict__ pfx, const int32_t* __restrict__ boff, int y,
                                                int n, int ntiles, int idx)   // inclusive count up to idx
{
    return pfx[(int64_t)y * n + idx] + boff[y * ntiles + idx / kShardTile];
}

// One workgroup: tile totals -> exclusive tile offsets (in place), then the per-peer plan.
// plan (int32, device, visible to the host): [0] anything moves (same on every rank) | send_cnt[world] |
// recv_cnt[world] | send_base[world];   rplan (device only): roff[world] | rbase[world]
__global__ __launch_bounds__(kBlock) void shard_plan_kernel(const int32_t* __restrict__ first_all, int64_t n_total, int n,
                                                            int rank, int world, const int32_t* __restrict__ pfx,
                                                            int32_t* __restrict__ boff, int ntiles,
                                                            int32_t* __restrict__ plan, int32_t* __restrict__ rplan,
                                                            int32_t* __restrict__ host_plan,
                                                            uint32_t* __restrict__ host_flag, uint32_t seq, int recv_cap,
                                                            int32_t* __restrict__ host_heads)
{
    __shared__ int32_t s_part[kBlock];
    for (int y = 0; y < 2; ++y) {   // exclusive scan of the tile totals, kBlock-sized chunks with a running carry
        int32_t carry = 0;
        for (int c0 = 0; c0 < ntiles; c0 += kBlock) {
            const int t = c0 + threadIdx.x;
            const int32_t v = t < ntiles ? boff[y * ntiles + t] : 0;
            s_part[threadIdx.x] = v;
            __syncthreads();
            for (int o = 1; o < kBlock; o <<= 1) {
                const int32_t add = (int)threadIdx.x >= o ? s_part[threadIdx.x - o] : 0;
                __syncthreads();
                s_part[threadIdx.x] += add;
                __syncthreads();
            }
            if (t < ntiles) boff[y * ntiles + t] = carry + s_part[threadIdx.x] - v;
            carry += s_part[kBlock - 1];
            __syncthreads();
        }
    }
    __threadfence_block();
    __syncthreads();
    // one thread per peer (their binary searches run side by side: one thread doing all peers in turn is ~10 us of
    // dependent L2 round trips per peer), then thread 0 strings the receive offsets together
    __shared__ int32_t s_scnt[kMaxRanks], s_rcnt[kMaxRanks], s_sbase[kMaxRanks], s_rbase[kMaxRanks], s_any[kMaxRanks];
    const int64_t my_lo = (int64_t)rank * n, my_hi = my_lo + n;
    if ((int)threadIdx.x < world) {
        const int q = threadIdx.x;
        const int64_t a_me = first_or_total(first_all, my_lo, n_total), b_me = first_or_total(first_all, my_hi, n_total);
        const int64_t aq = first_or_total(first_all, (int64_t)q * n, n_total);
        const int64_t bq = first_or_total(first_all, (int64_t)(q + 1) * n, n_total);
        s_any[q] = (q > 0 && aq != (int64_t)q * n) ? 1 : 0;   // a run boundary off a rank boundary: somebody exchanges
        {   // could rank q's staging area overflow?  Decided from the boundary values alone, so that EVERY rank reaches
            // the same verdict about EVERY rank: the slots of q with an ancestor on another rank bound what q receives
            const int64_t q0 = (int64_t)q * n, q1 = q0 + n;
            const int64_t l = aq > q0 ? aq : q0, h = bq < q1 ? bq : q1;
            const int64_t local_slots = h > l ? h - l : 0;
            if ((int64_t)n - local_slots > (int64_t)recv_cap) s_any[q] |= 2;
        }
        // what I receive from q: my slots [lo, hi) descend from q's particles
        int64_t lo = aq > my_lo ? aq : my_lo, hi = bq < my_hi ? bq : my_hi;
        int32_t rcnt = 0, rbase = 0;
        if (q != rank && hi > lo) {
            rbase = shard_prefix(pfx, boff, 0, n, ntiles, (int)(lo - my_lo));
            rcnt = shard_prefix(pfx, boff, 0, n, ntiles, (int)(hi - 1 - my_lo)) - rbase + 1;
        }
        // what I send to q: q's slots [lo, hi) descend from my particles
        const int64_t q_lo = (int64_t)q * n, q_hi = q_lo + n;
        lo = a_me > q_lo ? a_me : q_lo;
        hi = b_me < q_hi ? b_me : q_hi;
        int32_t scnt = 0, sbase = 0;
        if (q != rank && hi > lo) {
            const int a_lo = (int)(last_with_first_le(first_all, n_total, lo) - my_lo);
            const int a_hi = (int)(last_with_first_le(first_all, n_total, hi - 1) - my_lo);
            sbase = shard_prefix(pfx, boff, 1, n, ntiles, a_lo);
            scnt = shard_prefix(pfx, boff, 1, n, ntiles, a_hi) - sbase + 1;
        }
        s_scnt[q] = scnt;
        s_rcnt[q] = rcnt;
        s_sbase[q] = sbase;
        s_rbase[q] = rbase;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    int32_t anything = 0, roff = 0;
    for (int q = 0; q < world; ++q) {
        anything |= s_any[q];
        rplan[q] = roff;
        rplan[world + q] = s_rbase[q];
        roff += s_rcnt[q];
        plan[1 + q] = s_scnt[q];
        plan[1 + world + q] = s_rcnt[q];
        plan[1 + 2 * world + q] = s_sbase[q];
        if (host_plan) {
            host_plan[1 + q] = s_scnt[q];
            host_plan[1 + world + q] = s_rcnt[q];
            host_plan[1 + 2 * world + q] = s_sbase[q];
        }
    }
    if (host_heads) {   // distinct ancestors among my slots = heads (feedback for the choice of the EKF form)
        host_heads[0] = shard_prefix(pfx, boff, 0, n, ntiles, n - 1);
        host_heads[1] = n;
    }
    plan[0] = anything;   // bit 0: somebody exchanges rows; bit 1: some rank's staging area might not hold them
    if (host_plan) {   // zero-copy delivery: the host polls the flag instead of a device-to-host copy + stream sync
        host_plan[0] = anything;
        __threadfence_system();
        __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// src[jl]: where slot r*n + jl finds its ancestor — a local particle index, or n + its row in the staging tail
__global__ __launch_bounds__(kBlock) void ancestors_sharded_kernel(const int32_t* __restrict__ gsrc,
                                                                   const int32_t* __restrict__ pfx,
                                                                   const int32_t* __restrict__ boff, int ntiles,
                                                                   const int32_t* __restrict__ rplan, int n, int rank,
                                                                   int world, int32_t* __restrict__ src,
                                                                   int32_t* __restrict__ pose_idx)
{
    const int jl = blockIdx.x * kBlock + threadIdx.x;
    if (jl >= n) return;
    const int32_t g = gsrc[jl];
    const int owner = g / n;
    // where the ancestor's pose sits in an all-gather of the ranks' [x | y | theta] blocks (3 n floats per rank)
    if (pose_idx) pose_idx[jl] = owner * 3 * n + (g - owner * n);
    src[jl] = owner == rank ? g - rank * n
                            : n + rplan[owner] + (shard_prefix(pfx, boff, 0, n, ntiles, jl) - rplan[world + owner]);
}

// Pack what the other ranks need from me into one buffer: block d (for rank d) is cnt_d records of 3 + 5L floats
// — x, y, theta, then the five map planes (L values each) — one record per DISTINCT particle of mine with
// offspring among d's slots, in particle order.  One workgroup per record, one launch for every destination.
__global__ __launch_bounds__(kBlock) void migrate_pack_kernel(const int32_t* __restrict__ pfx,
                                                              const int32_t* __restrict__ boff, int ntiles, int n,
                                                              MigratePlan plan, const float* __restrict__ pose,
                                                              int64_t pose_ld, const float* __restrict__ map,
                                                              int64_t row_stride, int plane_stride, int nlandmarks,
                                                              float* __restrict__ out, const int32_t* __restrict__ pt,
                                                              int nb, const float* __restrict__ split_cov,
                                                              const int32_t* __restrict__ split_cls, PageGeom geom)
{
    const int p = blockIdx.x;
    int d = 0;
    while (p >= plan.off[d + 1]) ++d;
    const int32_t target = (int32_t)plan.lo[d] + (p - plan.off[d]);   // P value of the wanted particle
    int lo = 0, hi = n - 1;                                            // first a with P(a) >= target
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (shard_prefix(pfx, boff, 1, n, ntiles, mid) >= target) hi = mid; else lo = mid + 1;
    }
    const int loc = lo;
    float* __restrict__ rec = out + (int64_t)(3 + 5 * nlandmarks) * p;
    if (threadIdx.x < 3) rec[threadIdx.x] = pose[threadIdx.x * pose_ld + loc];
    if (pt && split_cls) {   // split pages: the means behind the page table (pages of two planes, kernels.h: PageGeom), the
        const int32_t* __restrict__ tab = pt + (int64_t)loc * nb;   // covariance planes in the class's rows
        const float* __restrict__ crow = split_cov + (int64_t)split_cls[loc] * 3 * plane_stride;
        for (int pl = 0; pl < 5; ++pl)
            for (int l = threadIdx.x; l < nlandmarks; l += kBlock) {
                const int64_t page = tab[l / kPageLandmarks];
                rec[3 + pl * nlandmarks + l] =
                    pl < 2 ? map[page * (2 * kPageLandmarks) + (page >= geom.half_pages ? geom.gap : 0) + pl * kPageLandmarks + l % kPageLandmarks]
                           : crow[(pl - 2) * plane_stride + l];
            }
        return;
    }
    if (pt) {   // paged maps: `map` is the page pool, the particle's landmarks sit behind its page table (paged_kernels.hip)
        const int32_t* __restrict__ tab = pt + (int64_t)loc * nb;
        for (int pl = 0; pl < 5; ++pl)
            for (int l = threadIdx.x; l < nlandmarks; l += kBlock)
                rec[3 + pl * nlandmarks + l] = map[(int64_t)tab[l / kPageLandmarks] * (5 * kPageLandmarks) + pl * kPageLandmarks +
                                                   l % kPageLandmarks];
        return;
    }
    const float* __restrict__ row = map + (int64_t)loc * row_stride;
    if (split_cls) {   // split layout: `map` holds the means (two planes), the covariance planes are the class's (split_kernels.hip)
        const float* __restrict__ crow = split_cov + (int64_t)split_cls[loc] * 3 * plane_stride;
        for (int pl = 0; pl < 5; ++pl)
            for (int l = threadIdx.x; l < nlandmarks; l += kBlock)
                rec[3 + pl * nlandmarks + l] = pl < 2 ? row[pl * plane_stride + l] : crow[(pl - 2) * plane_stride + l];
        return;
    }
    for (int pl = 0; pl < 5; ++pl)
        for (int l = threadIdx.x; l < nlandmarks; l += kBlock) rec[3 + pl * nlandmarks + l] = row[pl * plane_stride + l];
}

// Unpack the received records into the staging tail behind the n local particles (position = running index
// over sources in rank order, matching ancestors_sharded_kernel).
__global__ __launch_bounds__(kBlock) void migrate_unpack_kernel(const float* __restrict__ in, int total, int n,
                                                                float* __restrict__ pose, int64_t pose_ld,
                                                                float* __restrict__ map, int64_t row_stride,
                                                                int plane_stride, int nlandmarks)
{
    const int p = blockIdx.x;
    if (p >= total) return;
    const float* __restrict__ rec = in + (int64_t)(3 + 5 * nlandmarks) * p;
    if (threadIdx.x < 3) pose[threadIdx.x * pose_ld + n + p] = rec[threadIdx.x];
    float* __restrict__ row = map + (int64_t)(n + p) * row_stride;
    for (int pl = 0; pl < 5; ++pl)
        for (int l = threadIdx.x; l < nlandmarks; l += kBlock) row[pl * plane_stride + l] = rec[3 + pl * nlandmarks + l];
}

__global__ __launch_bounds__(kBlock) void gather_f32_kernel(const float* __restrict__ src,
                                                            const int32_t* __restrict__ idx, int n,
                                                            float* __restrict__ dst)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// out row i = in row idx[i]; one workgroup per particle
__global__ __launch_bounds__(kBlock) void gather_map_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int64_t in_row_stride, int64_t out_row_stride,
                                                            int in_plane_stride, int out_plane_stride,
                                                            int nlandmarks, const int32_t* __restrict__ idx, int n)
{
    const int i = blockIdx.x;
    if (i >= n) return;
    const float* __restrict__ src = in + (int64_t)idx[i] * in_row_stride;
    float* __restrict__ dst = out + (int64_t)i * out_row_stride;
    for (int pl = 0; pl < 5; ++pl)
        for (int l = threadIdx.x; l < nlandmarks; l += kBlock) dst[pl * out_plane_stride + l] = src[pl * in_plane_stride + l];
}

// index of the largest value, lowest index on ties (the heaviest particle); one workgroup
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ v, int n, int32_t* __restrict__ idx_out,
                                                      float* __restrict__ val_out)
{
    __shared__ float s_v[16];
    __shared__ int s_i[16];
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float x = v[i];
        if (x > bv || (x == bv && i < bi)) { bv = x; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = bv; s_i[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) { bv = s_v[w]; bi = s_i[w]; }
        *idx_out = bi == 0x7fffffff ? 0 : bi;
        *val_out = bv;
    }
}

// The heaviest particle with its pose, in one launch: {logw, global id, x, y, theta} to device memory and, optionally,
// to mapped host memory behind a sequence number (a plain C host then needs no copy and no stream synchronisation).
__global__ __launch_bounds__(1024) void best_particle_kernel(const float* __restrict__ v, int n,
                                                             const float* __restrict__ px, const float* __restrict__ py,
                                                             const float* __restrict__ pth, int64_t first_id,
                                                             float* __restrict__ out5, float* __restrict__ h_out5,
                                                             uint32_t* __restrict__ h_seq, uint32_t seq)
{
    __shared__ float s_v[16];
    __shared__ int s_i[16];
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float x = v[i];
        if (x > bv || (x == bv && i < bi)) { bv = x; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = bv; s_i[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int w = 1; w < 16; ++w)
        if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) { bv = s_v[w]; bi = s_i[w]; }
    if (bi == 0x7fffffff) bi = 0;
    const float r[5] = { bv, __int_as_float((int32_t)(first_id + bi)), px[bi], py[bi], pth[bi] };
    for (int k = 0; k < 5; ++k) out5[k] = r[k];
    if (h_out5) {
        for (int k = 0; k < 5; ++k) h_out5[k] = r[k];
        __threadfence_system();
        __hip_atomic_store(h_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Exact, order-independent sums over the population for the posterior mean: x and y as 2^-32 fixed point, the heading
// as sin / cos of (theta - ref) in 2^-30 fixed point (det_sincosf: the specified polynomial), accumulated with 64-bit
// integer atomics — the same bits for any summation order, workgroup count or sharding.  idx (optional): the pending
// resample gather.  The last workgroup to finish (a ticket) hands the four sums over — to device memory and,
// optionally, mapped host memory behind a sequence number — and clears the accumulators for the next call.
__global__ __launch_bounds__(kBlock) void pose_sums_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ th, const int32_t* __restrict__ idx,
                                                           int n, float ref_th, unsigned long long* __restrict__ acc,
                                                           unsigned int* __restrict__ ticket, long long* __restrict__ out4,
                                                           long long* __restrict__ h_out4, uint32_t* __restrict__ h_seq,
                                                           uint32_t seq)
{
    __shared__ uint64_t s_red[kBlock / 64];
    uint64_t sx = 0, sy = 0, ss = 0, sc = 0;   // two's complement: signed sums through unsigned adds
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const int j = idx ? idx[i] : i;
        float s, c;
        det_sincosf(th[j] - ref_th, s, c);
        sx += (uint64_t)(long long)(x[j] * 4294967296.0f);
        sy += (uint64_t)(long long)(y[j] * 4294967296.0f);
        ss += (uint64_t)(long long)(s * 1073741824.0f);
        sc += (uint64_t)(long long)(c * 1073741824.0f);
    }
    sx = block_sum_u64(sx, s_red);
    sy = block_sum_u64(sy, s_red);
    ss = block_sum_u64(ss, s_red);
    sc = block_sum_u64(sc, s_red);
    if (threadIdx.x != 0) return;
    atomicAdd(&acc[0], (unsigned long long)sx);
    atomicAdd(&acc[1], (unsigned long long)sy);
    atomicAdd(&acc[2], (unsigned long long)ss);
    atomicAdd(&acc[3], (unsigned long long)sc);
    __threadfence();
    if (atomicAdd(ticket, 1u) != gridDim.x - 1) return;
    __threadfence();
    long long r[4];
    for (int k = 0; k < 4; ++k) r[k] = (long long)atomicExch(&acc[k], 0ull);   // read and clear for the next call
    *ticket = 0;
    for (int k = 0; k < 4; ++k) out4[k] = r[k];
    if (h_out4) {
        for (int k = 0; k < 4; ++k) h_out4[k] = r[k];
        __threadfence_system();
        __hip_atomic_store(h_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

inline int blocks_for(int n) { return (n + kBlock - 1) / kBlock; }

}  // namespace

hipError_t launch_motion_sample(hipStream_t stream, const float* sx, const float* sy, const float* sth,
                                const int32_t* anc, float* x, float* y, float* th, int n, int64_t first_id,
                                const float dp[3], const float sigma[3], uint64_t seed, uint32_t frame)
{
    if (n <= 0) return hipSuccess;
    const MotionParams mp = make_motion_params(first_id, dp, sigma, seed, frame);
    motion_sample_kernel<<<blocks_for(n), kBlock, 0, stream>>>(sx, sy, sth, anc, x, y, th, n, mp);
    return hipGetLastError();
}

hipError_t launch_ekf_update(hipStream_t stream, const EkfArgs& a_in, const EventPair* ev, int group_size)
{
    if (a_in.n <= 0) return hipSuccess;
    EkfArgs a = a_in;
    int blocks = (a.n + kEkfWaves - 1) / kEkfWaves;
    a.xcd_chunk = 0;
    if (blocks >= 64) {   // XCD-contiguous numbering: pad the grid to a multiple of 8 (surplus workgroups exit at once)
        a.xcd_chunk = (blocks + 7) / 8;
        blocks = 8 * a.xcd_chunk;
    }
    if (a.cov) {   // split layout: always the grouped form (2 particles per wavefront unless the caller asks for 4 or 8)
        static const int forced = getenv("SLAM_SPLIT_G") ? atoi(getenv("SLAM_SPLIT_G")) : 0;   // measurements
        const int G = forced ? forced : (group_size == 4 || group_size == 8 ? group_size : 2);
        int gblocks = (a.n + kEkfWaves * G - 1) / (kEkfWaves * G);
        a.xcd_chunk = 0;
        if (gblocks >= 64) {
            a.xcd_chunk = (gblocks + 7) / 8;
            gblocks = 8 * a.xcd_chunk;
        }
        if (ev) (void)hipEventRecord(ev->start, stream);
        if (G == 8) ekf_split_kernel<EKF_SPLIT_NB, 8><<<gblocks, kEkfWaves * 64, 0, stream>>>(a);
        else if (G == 4) ekf_split_kernel<EKF_SPLIT_NB, 4><<<gblocks, kEkfWaves * 64, 0, stream>>>(a);
        else ekf_split_kernel<EKF_SPLIT_NB, 2><<<gblocks, kEkfWaves * 64, 0, stream>>>(a);
        if (ev) (void)hipEventRecord(ev->stop, stream);
        return hipGetLastError();
    }
    const bool copy = a.map_in != a.map_out;   // in place: rows without an observation stay as they are
    // out of place, more than one batch per row: optionally the grouped form (group_size neighbouring particles per
    // wavefront, shared source rows stay in registers); the caller knows roughly how many distinct ancestors the last
    // resample left (slam_ekf_form_set forces one form).
    if (copy && a.nlandmarks > 128 && group_size > 0) {
        // group size: measured on MI355X (64k x 500 | 1M x 1000 | 64k x 500 with 50 % distinct ancestors | 512k x 5000; one
        // wavefront per particle: 156 us | 4.28 ms | 177 us | 10.09 ms): 2 particles 148 | 4.09 | 169 | 9.79; 3: 135;
        // 4: 139 | 3.86 | 180 | 9.82; 6: 141; 8: 150 | 3.84 | 199 | 9.92.  Hence 4 when neighbours share ancestors, 2 when
        // they rarely do.  SLAM_EKF_G overrides (measurements).  Batches in flight per pass (the first template argument),
        // group of 4, 64k x 500 | 1M x 1000 | 512k x 5000: 1: 150 us | 3.97 ms; 2: 140-145 | 3.90-3.92 | 9.79; 3: 144 | 4.02;
        // 4: 137-139 | 3.86 | 9.86 — within the run-to-run spread: 2 kept (82 VGPRs, 5 waves per SIMD; 4 needs 114).
        static const int forced = getenv("SLAM_EKF_G") ? atoi(getenv("SLAM_EKF_G")) : 0;
        const int G = forced ? forced : group_size;
        int gblocks = (a.n + kEkfWaves * G - 1) / (kEkfWaves * G);
        a.xcd_chunk = 0;
        if (gblocks >= 64) {
            a.xcd_chunk = (gblocks + 7) / 8;
            gblocks = 8 * a.xcd_chunk;
        }
        if (ev) (void)hipEventRecord(ev->start, stream);
        switch (G) {
        case 2: ekf_update_group_kernel<EKF_GROUP_NB, 2><<<gblocks, kEkfWaves * 64, 0, stream>>>(a); break;
        case 8: ekf_update_group_kernel<EKF_GROUP_NB, 8><<<gblocks, kEkfWaves * 64, 0, stream>>>(a); break;
        default: ekf_update_group_kernel<EKF_GROUP_NB, 4><<<gblocks, kEkfWaves * 64, 0, stream>>>(a); break;
        }
        if (ev) (void)hipEventRecord(ev->stop, stream);
        return hipGetLastError();
    }
    // batches of 128 landmarks in flight per wavefront: 2 measured best at 64k x 500 (1: 178 us, 2: 166 us, 4: 180 us)
    const int nb = a.nlandmarks <= 128 ? 1 : 2;
    if (ev) (void)hipEventRecord(ev->start, stream);
    // in place: 4 batches (512 landmarks) per round trip; measured at 64k x 500 with 32 landmarks observed: 1 batch at a
    // time 91 us, because every batch is its own dependent chain obs table -> row -> store
    if (!copy && a.nlandmarks > 128) ekf_update_kernel<4, false><<<blocks, kEkfWaves * 64, 0, stream>>>(a);
    else if (!copy) ekf_update_kernel<1, false><<<blocks, kEkfWaves * 64, 0, stream>>>(a);
    else if (nb == 1) ekf_update_kernel<1, true><<<blocks, kEkfWaves * 64, 0, stream>>>(a);
    else ekf_update_kernel<2, true><<<blocks, kEkfWaves * 64, 0, stream>>>(a);
    if (ev) (void)hipEventRecord(ev->stop, stream);
    return hipGetLastError();
}

bool frame_front_fits(int n, int nlandmarks, int group_size)
{
    static const int wave_max = getenv("SLAM_SCORE_WAVE_MAX") ? atoi(getenv("SLAM_SCORE_WAVE_MAX")) : kWaveMaxPoses;
    if (n < wave_max || nlandmarks <= 128 || (group_size != 2 && group_size != 4 && group_size != 8)) return false;
    return (n + kEkfWaves * group_size - 1) / (kEkfWaves * group_size) >= 64;
}

// The front of a single-GPU frame in one launch (frame_front_kernel).  *launched = false when the shapes do not fit it (few
// particles: the one-wavefront-per-pose scorer; short rows; too few update workgroups for the XCD-contiguous numbering): the
// caller then issues the two launches.
hipError_t launch_frame_front(hipStream_t stream, const ScoreGrid& g, const float* bx, const float* by, int nbeams,
                              const MotionIO& io, int64_t first_id, const float dp[3], const float sigma[3], uint64_t seed,
                              uint32_t frame, float* score, int32_t* count, const EkfArgs& a_in, int group_size,
                              const EventPair* ev, bool* launched, int* lanes_per_pose)
{
    *launched = false;
    const int n = a_in.n;
    static const int quad_max = getenv("SLAM_SCORE_QUAD_MAX") ? atoi(getenv("SLAM_SCORE_QUAD_MAX")) : kQuadMaxPoses;
    static const int forced_g = getenv("SLAM_SPLIT_G") ? atoi(getenv("SLAM_SPLIT_G")) : 0;   // measurements (split layout)
    if (a_in.cov && forced_g) group_size = forced_g;
    if (!a_in.cov && group_size == 8) group_size = 4;   // rows: 2 or 4 particles per updating wavefront
    if (a_in.map_in == a_in.map_out || !frame_front_fits(n, a_in.nlandmarks, group_size)) return hipSuccess;
    const int G = group_size;
    const int gblocks = (n + kEkfWaves * G - 1) / (kEkfWaves * G);
    // inside the fused launch the scorer's latency is hidden anyway; what counts is how long its wavefronts hold slots the
    // update would fill: SLAM_FRONT_QUAD_MAX = population below which the 4-lanes-per-pose scorer is used here (measurements)
    static const int front_quad_max = getenv("SLAM_FRONT_QUAD_MAX") ? atoi(getenv("SLAM_FRONT_QUAD_MAX")) : quad_max;
    const bool quad = n < front_quad_max;
    FrontArgs f;
    f.g = g;
    f.bx = bx;
    f.by = by;
    f.nbeams = nbeams;
    f.score = score;
    f.count = count;
    f.mio = io;
    f.mpar = make_motion_params(first_id, dp, sigma, seed, frame);
    f.a = a_in;
    f.ekf_octets = (gblocks + 7) / 8;
    f.a.xcd_chunk = f.ekf_octets;
    f.score_blocks = (int)(((quad ? 4L : 1L) * n + kScoreBlock - 1) / kScoreBlock);
    f.score_octets = (f.score_blocks + 7) / 8;
    const int grid = 8 * (f.score_octets + f.ekf_octets);
    static const int span_pct = getenv("SLAM_FRONT_SPAN") ? atoi(getenv("SLAM_FRONT_SPAN")) : 100;
    const int64_t span = (int64_t)(f.score_octets + f.ekf_octets) * (span_pct < 1 ? 1 : span_pct > 100 ? 100 : span_pct) / 100;
    f.score_span = (int)(span > f.score_octets ? span : f.score_octets);
    const size_t lds = sizeof(float2) * (size_t)(nbeams + (quad ? 4 * kQuadDepth : kLaneDepth)) + (g.packed ? 1024 : 0);
    if (ev) (void)hipEventRecord(ev->start, stream);
    // the instantiation: particles per updating wavefront x scorer's lane mapping x map layout x grid copy the scorer reads
#define SLAM_FRONT(G_, LPP_, DEPTH_, SP_, PK_) \
    frame_front_kernel<(SP_) ? EKF_SPLIT_NB : EKF_GROUP_NB, G_, LPP_, DEPTH_, SP_, PK_><<<grid, kEkfWaves * 64, lds, stream>>>(f)
#define SLAM_FRONT_GL(SP_, PK_)                          \
    do {                                                 \
        if (quad) {                                      \
            if (G == 2) SLAM_FRONT(2, 4, kQuadDepth, SP_, PK_); \
            else if (G == 8 && (SP_)) SLAM_FRONT((SP_) ? 8 : 4, 4, kQuadDepth, SP_, PK_); \
            else SLAM_FRONT(4, 4, kQuadDepth, SP_, PK_);        \
        } else {                                         \
            if (G == 2) SLAM_FRONT(2, 1, kLaneDepth, SP_, PK_); \
            else if (G == 8 && (SP_)) SLAM_FRONT((SP_) ? 8 : 4, 1, kLaneDepth, SP_, PK_); \
            else SLAM_FRONT(4, 1, kLaneDepth, SP_, PK_);        \
        }                                                \
    } while (0)
    if (f.a.cov) {
        if (f.g.packed) SLAM_FRONT_GL(true, true);
        else SLAM_FRONT_GL(true, false);
    } else {
        if (f.g.packed) SLAM_FRONT_GL(false, true);
        else SLAM_FRONT_GL(false, false);
    }
#undef SLAM_FRONT_GL
#undef SLAM_FRONT
    if (ev) (void)hipEventRecord(ev->stop, stream);
    *launched = true;
    if (lanes_per_pose) *lanes_per_pose = quad ? 4 : 1;
    return hipGetLastError();
}

// every float whose exponent lies in the fast reciprocal's range (both signs): ekf_rcp_core against the compiler's IEEE
// division, the scalar form and the packed one; out[0] += mismatches, out[1] += values checked
namespace {
__global__ __launch_bounds__(256) void selftest_reciprocal_kernel(unsigned long long* __restrict__ out)
{
    unsigned long long bad = 0, seen = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x; b < (1ull << 32); b += (uint64_t)gridDim.x * 256) {
        const float d = __uint_as_float((uint32_t)b);
        if (!ekf_rcp_in_range(d)) continue;
        const float exact = 1.0f / d;
        const float fast = ekf_rcp_core(d);
        const v2f two = ekf_rcp((v2f){d, -d});   // (every lane here is in range: the packed fast path)
        bad += (__float_as_uint(exact) != __float_as_uint(fast)) || (__float_as_uint(two[0]) != __float_as_uint(exact)) ||
               (__float_as_uint(two[1]) != (__float_as_uint(exact) ^ 0x80000000u));
        ++seen;
    }
    if (bad) atomicAdd(&out[0], bad);
    atomicAdd(&out[1], seen);
}
}  // namespace

hipError_t launch_selftest_reciprocal(hipStream_t stream, unsigned long long* out)
{
    selftest_reciprocal_kernel<<<256 * 16, 256, 0, stream>>>(out);
    return hipGetLastError();
}

constexpr int kMaxWeightBlocks = 2048;
static int capped_blocks(int n) { const int b = blocks_for(n); return b < kMaxWeightBlocks ? b : kMaxWeightBlocks; }
int logweight_scratch_elems(int n) { return capped_blocks(n > 0 ? n : 1); }
int logweight_scratch_floats() { return kMaxWeightBlocks + 2; }   // block maxima + {ticket, running maximum} (zero-initialise once)

hipError_t launch_build_obs_list(hipStream_t stream, const float* tzx, const float* tzy, int L, int32_t* id, float* zx,
                                 float* zy, int32_t* round, int32_t* count, int32_t* h_count)
{
    build_obs_list_kernel<<<1, 1024, 0, stream>>>(tzx, tzy, L, id, zx, zy, round, count, h_count);
    return hipGetLastError();
}

hipError_t launch_ekf_sparse(hipStream_t stream, const EkfArgs& a_in, const int32_t* id, const float* zx, const float* zy,
                             const int32_t* round, const int32_t* count, const EventPair* ev)
{
    if (a_in.n <= 0) return hipSuccess;
    EkfArgs a = a_in;
    int blocks = (a.n + kEkfWaves - 1) / kEkfWaves;
    a.xcd_chunk = 0;
    if (blocks >= 64) {
        a.xcd_chunk = (blocks + 7) / 8;
        blocks = 8 * a.xcd_chunk;
    }
    ObsList ol{ id, zx, zy, round, count };
    if (ev) (void)hipEventRecord(ev->start, stream);
    ekf_sparse_kernel<<<blocks, kEkfWaves * 64, 0, stream>>>(a, ol);
    if (ev) (void)hipEventRecord(ev->stop, stream);
    return hipGetLastError();
}

hipError_t launch_logweight(hipStream_t stream, const float* score, const float* loglik, float gain, int n,
                            float* logw, float* block_max_scratch, float* d_max, const float* carry,
                            const int32_t* prev_resampled, const CovArgs* cov, int cov_bound)
{
    if (n <= 0) return hipSuccess;
    const int nb = capped_blocks(n);
    if (cov && cov_bound > 0) {   // + the covariance classes' update (launch_cov_update's grid, flattened)
        const int ly = cov->nlandmarks > 0 ? (cov->nlandmarks + 255) / 256 : 1;
        logweight_cov_kernel<<<nb + (int64_t)cov_bound * ly, kBlock, 0, stream>>>(score, loglik, gain, n, carry, prev_resampled, logw,
                                                                               block_max_scratch, nb, ly, *cov);
    } else
        logweight_kernel<<<nb, kBlock, 0, stream>>>(score, loglik, gain, n, carry, prev_resampled, logw, block_max_scratch);
    if (d_max) max_finalize_kernel<<<1, kBlock, 0, stream>>>(block_max_scratch, nb, d_max);
    return hipGetLastError();
}

hipError_t launch_quantise_weights(hipStream_t stream, const float* logw, const float* d_max, int n, uint64_t* wq,
                                   uint64_t* d_sum)
{
    hipError_t err = hipMemsetAsync(d_sum, 0, sizeof(uint64_t), stream);
    if (err != hipSuccess) return err;
    if (n <= 0) return hipSuccess;
    quantise_weights_kernel<<<capped_blocks(n), kBlock, 0, stream>>>(logw, d_max, n, wq,
                                                                     reinterpret_cast<unsigned long long*>(d_sum));
    return hipGetLastError();
}

int prefix_sum_scratch_elems(int n) { return (n + kScanTile - 1) / kScanTile + 1; }
int scan_tile_count(int n) { return (n + kScanTile - 1) / kScanTile; }

hipError_t launch_prefix_sum(hipStream_t stream, const uint64_t* in, int n, uint64_t* out, uint64_t* block_scratch)
{
    if (n <= 0) return hipSuccess;
    const int ntiles = (n + kScanTile - 1) / kScanTile;
    scan_tiles_kernel<<<ntiles, kBlock, 0, stream>>>(in, n, out, block_scratch);
    if (ntiles > 1) {
        scan_totals_kernel<<<1, kBlock, 0, stream>>>(block_scratch, ntiles);
        add_tile_offsets_kernel<<<ntiles, kBlock, 0, stream>>>(out, n, block_scratch);
    }
    return hipGetLastError();
}

hipError_t launch_quantise_scan(hipStream_t stream, const float* logw, const float* d_max, const float* block_max,
                                int nblock_max, int n, uint64_t* cdf_local, uint64_t* tile_total, uint64_t* d_sum,
                                float* carry, uint64_t* tile_s16, uint64_t* tile_q16, unsigned int* ticket)
{
    if (n <= 0) return hipSuccess;
    const int ntiles = (n + kScanTile - 1) / kScanTile;
    // `ticket`: {ticket, pad, three 64-bit accumulators} (8-byte aligned behind the pad), zero-initialised, left zeroed
    if (carry)
        quantise_scan_kernel<true><<<ntiles, kBlock, 0, stream>>>(logw, d_max, block_max, nblock_max, n, cdf_local, tile_total,
                                                                  carry, tile_s16, tile_q16, d_sum, ticket);
    else
        quantise_scan_kernel<false><<<ntiles, kBlock, 0, stream>>>(logw, d_max, block_max, nblock_max, n, cdf_local,
                                                                   tile_total, nullptr, nullptr, nullptr, d_sum, ticket);
    return hipGetLastError();
}

hipError_t launch_offspring_from_scan(hipStream_t stream, const uint64_t* cdf_local, const uint64_t* tile_total, int n,
                                      const uint64_t* d_base, const uint64_t* d_total, const uint64_t* d_shard_totals,
                                      int rank, int world, uint64_t seed, uint32_t frame, int64_t n_total,
                                      int32_t* first, uint32_t frac_q16, const GateOut& gate)
{
    if (n <= 0) return hipSuccess;
    const int ntiles = (n + kScanTile - 1) / kScanTile;
    offspring_from_scan_kernel<<<blocks_for(n), kBlock, 0, stream>>>(cdf_local, tile_total, ntiles, n, d_base, d_total,
                                                                     d_shard_totals, rank, world, (uint32_t)seed,
                                                                     (uint32_t)(seed >> 32), frame, (uint64_t)n_total,
                                                                     first, tile_total + ntiles, tile_total + 2 * ntiles,
                                                                     frac_q16, gate);
    return hipGetLastError();
}

hipError_t launch_offspring_offsets(hipStream_t stream, const uint64_t* cdf, int n, const uint64_t* d_base,
                                    const uint64_t* d_total, uint64_t seed, uint32_t frame, int64_t n_total,
                                    int32_t* first)
{
    if (n <= 0) return hipSuccess;
    offspring_offsets_kernel<<<blocks_for(n), kBlock, 0, stream>>>(cdf, n, d_base, d_total, (uint32_t)seed,
                                                                   (uint32_t)(seed >> 32), frame, (uint64_t)n_total,
                                                                   first);
    return hipGetLastError();
}

bool ancestors_from_scan_fits(int n) { return n > 0 && (n + kScanTile - 1) / kScanTile <= kMaxLdsTiles; }

hipError_t launch_ancestors_from_scan(hipStream_t stream, const uint64_t* cdf_local, const uint64_t* tile_total, int n,
                                      uint64_t seed, uint32_t frame, int32_t* anc, uint32_t frac_q16, const GateOut& gate,
                                      const HeadsOut& heads)
{
    if (n <= 0) return hipSuccess;
    const int ntiles = (n + kScanTile - 1) / kScanTile;
    const uint64_t *ts = tile_total + ntiles, *tq = tile_total + 2 * ntiles;   // layout of the engine's scan state
    if (ntiles <= kBlock)
        ancestors_from_scan_kernel<1><<<blocks_for(n), kBlock, 0, stream>>>(cdf_local, tile_total, ntiles, n, (uint32_t)seed,
                                                                            (uint32_t)(seed >> 32), frame, anc, ts, tq,
                                                                            frac_q16, gate, heads);
    else
        ancestors_from_scan_kernel<kMaxLdsTiles / kBlock><<<blocks_for(n), kBlock, 0, stream>>>(
            cdf_local, tile_total, ntiles, n, (uint32_t)seed, (uint32_t)(seed >> 32), frame, anc, ts, tq, frac_q16, gate,
            heads);
    return hipGetLastError();
}

hipError_t launch_ancestors(hipStream_t stream, const int32_t* first_all, int64_t n_total, int64_t slot0, int nslots,
                            int32_t* anc)
{
    if (nslots <= 0) return hipSuccess;
    ancestors_kernel<<<blocks_for(nslots), kBlock, 0, stream>>>(first_all, n_total, slot0, nslots, anc);
    return hipGetLastError();
}

int shard_scan_words(int n) { const int t = (n + kShardTile - 1) / kShardTile; return 3 * n + 2 * t + 2 * kMaxRanks; }

// scratch (int32 words, shard_scan_words(n)): gsrc[n] | pfx[2][n] | boff[2][ntiles] | rplan[2*kMaxRanks]
hipError_t launch_ancestors_sharded(hipStream_t stream, const int32_t* first_all, int64_t n_total, int n, int rank,
                                    int world, int32_t* scratch, int32_t* plan, int32_t* src, int32_t* pose_idx,
                                    int32_t* host_plan, uint32_t* host_flag, uint32_t seq, int recv_cap,
                                    int32_t* host_heads)
{
    if (n <= 0) return hipSuccess;
    const int ntiles = (n + kShardTile - 1) / kShardTile;
    int32_t* gsrc = scratch;
    int32_t* pfx = gsrc + n;
    int32_t* boff = pfx + 2 * (int64_t)n;
    int32_t* rplan = boff + 2 * ntiles;
    shard_search_kernel<<<blocks_for(n), kBlock, 0, stream>>>(first_all, n_total, n, rank, gsrc);
    shard_flag_scan_kernel<<<dim3(ntiles, 2), kBlock, 0, stream>>>(first_all, n_total, n, rank, gsrc, pfx, boff, ntiles);
    shard_plan_kernel<<<1, kBlock, 0, stream>>>(first_all, n_total, n, rank, world, pfx, boff, ntiles, plan, rplan,
                                                host_plan, host_flag, seq, recv_cap, host_heads);
    ancestors_sharded_kernel<<<blocks_for(n), kBlock, 0, stream>>>(gsrc, pfx, boff, ntiles, rplan, n, rank, world, src,
                                                                   pose_idx);
    return hipGetLastError();
}

// plan.lo[d] = send_base[d] (the P value of the first particle sent to d), plan.off = running record offsets
hipError_t launch_migrate_pack(hipStream_t stream, const int32_t* scratch, int n, const MigratePlan& plan,
                               const float* pose, int64_t pose_ld, const float* map, int64_t row_stride,
                               int plane_stride, int nlandmarks, float* out, const int32_t* pt, int nb, const float* split_cov,
                               const int32_t* split_cls, const PageGeom& geom)
{
    const int total = plan.off[plan.world];
    if (total <= 0) return hipSuccess;
    const int ntiles = (n + kShardTile - 1) / kShardTile;
    const int32_t* pfx = scratch + n;
    const int32_t* boff = pfx + 2 * (int64_t)n;
    migrate_pack_kernel<<<total, kBlock, 0, stream>>>(pfx, boff, ntiles, n, plan, pose, pose_ld, map, row_stride,
                                                     plane_stride, nlandmarks, out, pt, nb, split_cov, split_cls, geom);
    return hipGetLastError();
}

hipError_t launch_migrate_unpack(hipStream_t stream, const float* in, const MigratePlan& plan, int n, float* pose,
                                 int64_t pose_ld, float* map, int64_t row_stride, int plane_stride, int nlandmarks)
{
    const int total = plan.off[plan.world];
    if (total <= 0) return hipSuccess;
    migrate_unpack_kernel<<<total, kBlock, 0, stream>>>(in, total, n, pose, pose_ld, map, row_stride, plane_stride,
                                                       nlandmarks);
    return hipGetLastError();
}

hipError_t launch_argmax(hipStream_t stream, const float* v, int n, int32_t* idx_out, float* val_out)
{
    if (n <= 0) return hipSuccess;
    argmax_kernel<<<1, 1024, 0, stream>>>(v, n, idx_out, val_out);
    return hipGetLastError();
}

hipError_t launch_best_particle(hipStream_t stream, const float* v, int n, const float* px, const float* py,
                                const float* pth, int64_t first_id, float* out5, float* h_out5, uint32_t* h_seq,
                                uint32_t seq)
{
    if (n <= 0) return hipSuccess;
    best_particle_kernel<<<1, 1024, 0, stream>>>(v, n, px, py, pth, first_id, out5, h_out5, h_seq, seq);
    return hipGetLastError();
}

hipError_t launch_pose_sums(hipStream_t stream, const float* x, const float* y, const float* th, const int32_t* idx,
                            int n, float ref_th, unsigned long long* acc, unsigned int* ticket, long long* out4,
                            long long* h_out4, uint32_t* h_seq, uint32_t seq)
{
    if (n <= 0) return hipSuccess;
    const int nb = blocks_for(n) < 256 ? blocks_for(n) : 256;
    pose_sums_kernel<<<nb, kBlock, 0, stream>>>(x, y, th, idx, n, ref_th, acc, ticket, out4, h_out4, h_seq, seq);
    return hipGetLastError();
}

// ---- measurement support (slam_profile_copy_ceiling): the access shape of ekf_update_kernel without its arithmetic
__global__ __launch_bounds__(kEkfWaves * 64) void copy_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int n,
                                                                   int plane_stride, int xcd_chunk)
{
    const unsigned lane = threadIdx.x & 63u;
    const int wave = threadIdx.x >> 6;
    int bid = blockIdx.x;
    if (xcd_chunk > 0) bid = (bid & 7) * xcd_chunk + (bid >> 3);
    const int i = bid * kEkfWaves + wave;
    if (i >= n) return;
    const float* rin = in + (size_t)i * 5 * plane_stride;
    float* rout = out + (size_t)i * 5 * plane_stride;
    for (int lb = 0; lb < plane_stride; lb += 256) {   // two batches of 128 landmarks: every load before the first store
        const int nb = plane_stride - lb >= 256 ? 2 : 1;
        float m[2][2][5];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 5; ++p)
                    if (g < nb) m[g][t][p] = rin[p * plane_stride + lb + g * 128 + t * 64 + lane];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 5; ++p)
                    if (g < nb) __builtin_nontemporal_store(m[g][t][p], &rout[p * plane_stride + lb + g * 128 + t * 64 + lane]);
    }
}

hipError_t launch_copy_rows(hipStream_t stream, const float* in, float* out, int n, int plane_stride)
{
    const int blocks = (n + kEkfWaves - 1) / kEkfWaves, chunk = (blocks + 7) / 8;
    copy_rows_kernel<<<chunk * 8, kEkfWaves * 64, 0, stream>>>(in, out, n, plane_stride, chunk);
    return hipGetLastError();
}

hipError_t launch_gather_f32(hipStream_t stream, const float* src, const int32_t* idx, int n, float* dst)
{
    if (n <= 0) return hipSuccess;
    gather_f32_kernel<<<blocks_for(n), kBlock, 0, stream>>>(src, idx, n, dst);
    return hipGetLastError();
}

hipError_t launch_gather_map(hipStream_t stream, const float* in, float* out, int64_t in_row_stride,
                             int64_t out_row_stride, int in_plane_stride, int out_plane_stride, int nlandmarks,
                             const int32_t* idx, int n)
{
    if (n <= 0 || nlandmarks <= 0) return hipSuccess;
    gather_map_kernel<<<n, kBlock, 0, stream>>>(in, out, in_row_stride, out_row_stride, in_plane_stride,
                                                out_plane_stride, nlandmarks, idx, n);
    return hipGetLastError();
}

}  // namespace slam
