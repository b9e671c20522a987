// mapper.hip — slam_mapper_*: the reference's whole per-frame pipeline behind one call, with the map, the
// scan, the local map and both grids resident on the device (SURVEY.md §8f rows N1 + N2; N3's binary frames
// feed it directly).  The control flow mirrors Subsystem_1/main.c:844-969 step by step; only what the
// reference computes with libm (cos/sin of the beam angles, of the pose and of the lattice headings) and the
// arg-min / key-frame decisions stay on the host.  Results are bit-identical to the reference program.

#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <new>
#include <vector>

#include "engine_internal.h"

using namespace slam;

namespace {
enum { kMapCap = 20000, kLocalCap = 25000, kCoarseLd = 200, kFineLd = 400 };   // main.c:124, :148, :201, :207
}

struct slam_mapper {
    slam_engine* e = nullptr;
    int nbeams = 0;
    slam_mapper_params par{};   // the reference's run-time parameters (main.c:832-839, :50, :846, :224, :943)
    // device state
    DevBuf buf;   // one allocation, carved below
    float *d_cos = nullptr, *d_sin = nullptr, *d_range = nullptr, *d_bx = nullptr, *d_by = nullptr, *d_tx = nullptr,
          *d_ty = nullptr, *d_mx = nullptr, *d_my = nullptr, *d_lx = nullptr, *d_ly = nullptr, *d_hits = nullptr,
          *d_edt[2] = { nullptr, nullptr };
    int32_t *d_occ[2] = { nullptr, nullptr }, *d_counts = nullptr;   // counts: [0] nscan [1] msize [2] lsize
    slam_grid_meta* d_meta = nullptr;                                // [2]
    int map_cap = 0;
    // host state (the reference's locals of main())
    float pose[3] = { 0, 0, 0 }, prev[3] = { 0, 0, 0 }, map_pose[3] = { 0, 0, 0 };
    int mini_updated = 1, frame = 0;
    int32_t nhits = 0;
    float* h_range = nullptr;   // pinned and mapped: TWO frames of raw ranges, used in turn; the clean-up kernel reads them in place
    float* d_hrange = nullptr;  // ... as the device sees them
    int range_slot = 0;
};

namespace {

#define MP_TRY(call) SLAM_HIP_TRY(m->e, call)

int upload_ranges(slam_mapper* m, const float* ranges)
{
    hipStream_t st = m->e->stream;
    // Zero copy: the raw ranges go into pinned memory the clean-up kernel reads directly (one 4 KB pass; a host-to-device copy
    // command in front of it cost the frame's launch chain more than the kernel itself).  Two slots used in turn: the kernel
    // that read a slot two frames ago has finished — every frame ends with the host holding its matcher result, which is
    // behind that frame's clean-up in the stream — so no synchronisation is needed before the slot is written again.
    const size_t nb = (size_t)m->nbeams;
    float* h = m->h_range + (size_t)m->range_slot * nb;
    const float* d = m->d_hrange + (size_t)m->range_slot * nb;
    m->range_slot ^= 1;
    memcpy(h, ranges, sizeof(float) * nb);
    // main.c:863 readAScan(24): range_min 0.023 (main.c:50), usable range 24
    MP_TRY(launch_clean_scan(st, d, m->d_cos, m->d_sin, m->nbeams, m->par.range_min, m->par.usable_range, m->d_bx,
                             m->d_by, m->d_counts + 0));
    return SLAM_OK;
}

int to_world(slam_mapper* m, const float pose[3])
{
    // main.c:101-116 — cos/sin of the pose from libm on the host
    MP_TRY(launch_transform(m->e->stream, m->d_bx, m->d_by, m->d_counts + 0, pose[0], pose[1], cosf(pose[2]),
                            sinf(pose[2]), m->d_tx, m->d_ty));
    return SLAM_OK;
}

int rebuild_grids(slam_mapper* m)
{
    hipStream_t st = m->e->stream;
    // main.c:870-871: ExtractLocalMap(borderSize = 1), OccupationalGrid(0.2, 0.1)
    MP_TRY(launch_crop(st, m->d_tx, m->d_ty, m->d_counts + 0, m->par.border, m->d_mx, m->d_my, m->d_counts + 1, kLocalCap,
                       m->d_lx, m->d_ly, m->d_counts + 2));
    const float pix[2] = { m->par.pixel, m->par.pixel2 };
    const int ld[2] = { kCoarseLd, kFineLd };
    for (int k = 0; k < 2; ++k)
        MP_TRY(launch_rasterise(st, m->d_lx, m->d_ly, m->d_counts + 2, pix[k], ld[k], m->d_occ[k], m->d_meta + k));
    slam_grid_meta h_meta[2];
    MP_TRY(hipMemcpyAsync(h_meta, m->d_meta, sizeof h_meta, hipMemcpyDeviceToHost, st));
    MP_TRY(hipStreamSynchronize(st));   // key frames only: the EDT launch needs the grid extents
    for (int k = 0; k < 2; ++k) {
        if (h_meta[k].rows < 1 || h_meta[k].cols < 1 || h_meta[k].rows > ld[k] || h_meta[k].cols > ld[k])
            return SLAM_ERR_CAPACITY;   // the reference would overrun its 200^2 / 400^2 grids here
        MP_TRY(launch_edt(st, m->d_occ[k], ld[k], h_meta[k].rows, h_meta[k].cols, m->par.edt_cap /* main.c:224 */, m->d_edt[k]));
        int rc = slam_grid_set_dev(m->e, k, m->d_edt[k], &h_meta[k]);
        if (rc != SLAM_OK) return rc;
    }
    return SLAM_OK;
}

}  // namespace

extern "C" {

void slam_mapper_params_default(slam_mapper_params* p)
{
    if (!p) return;
    const slam_mapper_params d = { { 0.05f, 0.05f, 0.008727f },    // main.c:832
                                   { 0.025f, 0.025f, 0.004363f },  // main.c:833
                                   1.0f, 0.2f, 0.1f,               // main.c:834-836
                                   0.3f, 0.0872665f,               // main.c:838-839
                                   0.023f, 24.0f,                  // main.c:50, :846
                                   10.0f, 1.5f };                  // main.c:224, :943
    *p = d;
}

int slam_mapper_create(slam_engine* e, int nbeams, float angle_min, float angle_inc, slam_mapper** out)
{
    return slam_mapper_create_ex(e, nbeams, angle_min, angle_inc, nullptr, out);
}

int slam_mapper_create_ex(slam_engine* e, int nbeams, float angle_min, float angle_inc, const slam_mapper_params* params,
                          slam_mapper** out)
{
    if (!e || !out || nbeams <= 0) return SLAM_ERR_INVALID_ARG;
    if (nbeams > SLAM_MAX_BEAMS) return SLAM_ERR_CAPACITY;
    slam_mapper_params par;
    slam_mapper_params_default(&par);
    if (params) par = *params;
    if (!(par.pixel > 0.0f) || !(par.pixel2 > 0.0f) || !(par.border >= 0.0f) || !(par.edt_cap >= 0.0f) || !(par.key_dt >= 0.0f) ||
        !(par.key_dr >= 0.0f) || !(par.usable_range >= par.range_min))
        return SLAM_ERR_INVALID_ARG;
    if (ceilf(par.edt_cap) > (float)EDT_MAX_RADIUS) return SLAM_ERR_CAPACITY;
    *out = nullptr;
    if (int rc = slam_engine_sync(e)) return rc;
    slam_mapper* m = new (std::nothrow) slam_mapper();
    if (!m) return SLAM_ERR_HIP;
    m->e = e;
    m->nbeams = nbeams;
    m->par = par;
    m->map_cap = kMapCap + nbeams;
    const size_t nb = (size_t)nbeams;
    const size_t floats = 8 * nb + 2 * (size_t)m->map_cap + 2 * (size_t)kLocalCap + (size_t)kCoarseLd * kCoarseLd +
                          (size_t)kFineLd * kFineLd;
    const size_t ints = (size_t)kCoarseLd * kCoarseLd + (size_t)kFineLd * kFineLd + 16;
    if (m->buf.ensure(4 * (floats + ints) + 2 * sizeof(slam_grid_meta) + 64) != hipSuccess ||
        hipHostMalloc((void**)&m->h_range, 2 * 4 * nb, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&m->d_hrange, m->h_range, 0) != hipSuccess) {
        (void)hipGetLastError();
        slam_mapper_destroy(m);
        return SLAM_ERR_HIP;
    }
    if (hipMemset(m->buf.p, 0, m->buf.cap) != hipSuccess) {
        (void)hipGetLastError();
        slam_mapper_destroy(m);
        return SLAM_ERR_HIP;
    }
    float* f = m->buf.as<float>();
    auto take = [&](size_t n) { float* p = f; f += n; return p; };
    m->d_cos = take(nb); m->d_sin = take(nb); m->d_range = take(nb); m->d_bx = take(nb); m->d_by = take(nb);
    m->d_tx = take(nb); m->d_ty = take(nb); m->d_hits = take(nb);
    m->d_mx = take(m->map_cap); m->d_my = take(m->map_cap); m->d_lx = take(kLocalCap); m->d_ly = take(kLocalCap);
    m->d_edt[0] = take((size_t)kCoarseLd * kCoarseLd); m->d_edt[1] = take((size_t)kFineLd * kFineLd);
    int32_t* ip = reinterpret_cast<int32_t*>(f);
    m->d_occ[0] = ip; ip += (size_t)kCoarseLd * kCoarseLd;
    m->d_occ[1] = ip; ip += (size_t)kFineLd * kFineLd;
    m->d_counts = ip; ip += 16;
    m->d_meta = reinterpret_cast<slam_grid_meta*>(ip);
    // main.c:53-57 running-sum angle table, then cosf/sinf of every entry once (the reference recomputes the
    // same values every frame, main.c:89-90)
    std::vector<float> c(nb), s(nb);
    float a = angle_min;
    for (size_t k = 0; k < nb; ++k, a += angle_inc) {
        c[k] = cosf(a);
        s[k] = sinf(a);
    }
    if (hipMemcpy(m->d_cos, c.data(), 4 * nb, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->d_sin, s.data(), 4 * nb, hipMemcpyHostToDevice) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        slam_mapper_destroy(m);
        return SLAM_ERR_HIP;
    }
    *out = m;
    return SLAM_OK;
}

int slam_mapper_destroy(slam_mapper* m)
{
    if (!m) return SLAM_OK;
    (void)slam_engine_sync(m->e);
    m->buf.release();
    if (m->h_range) (void)hipHostFree(m->h_range);
    delete m;
    return SLAM_OK;
}

int slam_mapper_first_frame(slam_mapper* m, const float* ranges)
{
    if (!m || !ranges) return SLAM_ERR_INVALID_ARG;
    SLAM_HIP_TRY(m->e, hipSetDevice(m->e->device));
    // main.c:844-858: scan 0 at the origin seeds the map; the loop starts "mini-updated"
    const float origin[3] = { 0, 0, 0 };
    memcpy(m->pose, origin, sizeof origin);
    memcpy(m->prev, origin, sizeof origin);
    memcpy(m->map_pose, origin, sizeof origin);
    if (int rc = upload_ranges(m, ranges)) return rc;
    if (int rc = to_world(m, origin)) return rc;
    hipStream_t st = m->e->stream;
    MP_TRY(hipMemsetAsync(m->d_counts + 1, 0, 4, st));
    // Initialise (main.c:136-145): map := every world point of scan 0 = an append with every hit "new"
    MP_TRY(hipMemsetAsync(m->d_hits, 0x7f, 4 * (size_t)m->nbeams, st));   // 0x7f7f7f7f = 3.4e38 > 1.5
    int32_t nscan = 0;
    MP_TRY(hipMemcpyAsync(&nscan, m->d_counts + 0, 4, hipMemcpyDeviceToHost, st));
    MP_TRY(hipStreamSynchronize(st));
    MP_TRY(launch_map_append(st, m->d_hits, nscan, m->d_tx, m->d_ty, m->d_mx, m->d_my, m->d_counts + 1, m->map_cap, 1.5f));
    MP_TRY(hipMemsetAsync(m->d_hits, 0, 4 * (size_t)m->nbeams, st));
    m->mini_updated = 1;
    m->frame = 1;
    m->nhits = 0;
    return SLAM_OK;
}

int slam_mapper_next_frame(slam_mapper* m, const float* ranges, float pose_out[3])
{
    if (!m || !ranges || !pose_out || m->frame < 1) return SLAM_ERR_INVALID_ARG;
    SLAM_HIP_TRY(m->e, hipSetDevice(m->e->device));
    const float* coarse = m->par.fast_res;   // main.c:832
    const float* fine = m->par.fast_res2;    // main.c:833
    if (int rc = upload_ranges(m, ranges)) return rc;
    int in_world = 0;
    if (m->mini_updated) {   // main.c:865-872 — world points from the OLD pose
        if (int rc = to_world(m, m->pose)) return rc;
        in_world = 1;
        if (int rc = rebuild_grids(m)) return rc;
    }
    float guess[3];   // main.c:875-898
    for (int a = 0; a < 3; ++a) guess[a] = m->frame > 1 ? m->pose[a] + (m->pose[a] - m->prev[a]) : m->pose[a];
    // main.c:901-918 (coarse step on the fine grid when the map was not just rebuilt)
    // both calls as one round trip (SLAM_FASTMATCH_PAIR=0: one after the other, the host in between)
    static const bool pair = !(getenv("SLAM_FASTMATCH_PAIR") && atoi(getenv("SLAM_FASTMATCH_PAIR")) == 0);
    float m1[3], m2[3];
    int rc;
    if (pair) {
        rc = slam_engine_fastmatch_pair(m->e, m->mini_updated ? 0 : 1, 1, m->d_bx, m->d_by, m->nbeams, m->d_counts + 0, guess, coarse,
                                        fine, m2, &m->nhits, m->d_hits);
        if (rc != SLAM_OK) return rc;
    } else {
        rc = slam_engine_fastmatch(m->e, m->mini_updated ? 0 : 1, m->d_bx, m->d_by, m->nbeams, m->d_counts + 0, guess,
                                   coarse, m1, nullptr, &m->nhits, nullptr, m->d_hits);
        if (rc != SLAM_OK) return rc;
        rc = slam_engine_fastmatch(m->e, 1, m->d_bx, m->d_by, m->nbeams, m->d_counts + 0, m1, fine, m2, nullptr, &m->nhits,
                                   nullptr, m->d_hits);
        if (rc != SLAM_OK) return rc;
    }
    memcpy(m->prev, m->pose, sizeof m->prev);
    memcpy(m->pose, m2, sizeof m->pose);
    // main.c:928-961
    if (fabsf(m->pose[0] - m->map_pose[0]) > m->par.key_dt || fabsf(m->pose[1] - m->map_pose[1]) > m->par.key_dt ||
        fabsf(m->pose[2] - m->map_pose[2]) > m->par.key_dr) {
        m->mini_updated = 1;
        if (!in_world)
            if (int rc2 = to_world(m, m->pose)) return rc2;
        MP_TRY(launch_map_append(m->e->stream, m->d_hits, m->nhits, m->d_tx, m->d_ty, m->d_mx, m->d_my,
                                 m->d_counts + 1, m->map_cap, m->par.new_point_threshold));
        memcpy(m->map_pose, m->pose, sizeof m->map_pose);
    } else {
        m->mini_updated = 0;
    }
    m->frame++;
    memcpy(pose_out, m->pose, sizeof m->pose);
    return SLAM_OK;
}

int slam_mapper_get_map_host(slam_mapper* m, float* x, float* y, int32_t capacity, int32_t* n)
{
    if (!m || !n) return SLAM_ERR_INVALID_ARG;
    SLAM_HIP_TRY(m->e, hipSetDevice(m->e->device));
    if (int rc = slam_engine_sync(m->e)) return rc;
    int32_t sz = 0;
    MP_TRY(hipMemcpy(&sz, m->d_counts + 1, 4, hipMemcpyDeviceToHost));
    *n = sz;
    if (x && y) {
        const int32_t take = sz < capacity ? sz : capacity;
        if (take > 0) {
            MP_TRY(hipMemcpy(x, m->d_mx, 4 * (size_t)take, hipMemcpyDeviceToHost));
            MP_TRY(hipMemcpy(y, m->d_my, 4 * (size_t)take, hipMemcpyDeviceToHost));
        }
    }
    return SLAM_OK;
}

}  // extern "C"
