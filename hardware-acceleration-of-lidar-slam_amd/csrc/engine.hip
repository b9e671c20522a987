// engine.hip — implementation of the C ABI declared in include/slam_hip.h.
//
// The engine owns what the reference keeps in file-scope globals — the two occupancy/EDT grids
// (`occ_grid`, Subsystem_1/main.c:200-213), the current scan (`scan`, main.c:60-69) and the matcher
// result scratch (`FastMatchParameters`, main.c:374-379) — but as device-resident buffers behind an
// opaque handle, the shape of the `accel` handle of the reference's FPGA variant
// (Submodule_2/Hadrware_acclereated.cpp:842-845).  There is no CPU fallback anywhere in this file.

#include <hip/hip_runtime.h>
#include <type_traits>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <limits>
#include <vector>

#include "comm.h"
#include "engine_internal.h"

using namespace slam;

namespace {

int fail_hip(slam_engine* e, hipError_t err, const char* what)
{
    if (e) snprintf(e->err, sizeof e->err, "%s: %s", what, hipGetErrorString(err));
    (void)hipGetLastError();   // clear the sticky error so later calls can proceed
    return SLAM_ERR_HIP;
}

#define HIP_TRY(call)                                              \
    do {                                                           \
        hipError_t err__ = (call);                                 \
        if (err__ != hipSuccess) return fail_hip(e, err__, #call); \
    } while (0)

#define ENTER(e)                                 \
    do {                                         \
        if (!(e)) return SLAM_ERR_INVALID_ARG;   \
        HIP_TRY(hipSetDevice((e)->device));      \
    } while (0)

bool slot_ok(int slot) { return slot >= 0 && slot < SLAM_MAX_GRID_SLOTS; }

bool meta_ok(const slam_grid_meta* m)
{
    return m && m->rows >= 0 && m->cols >= 0 && m->ld >= m->cols && m->pixel > 0.0f &&
           (int64_t)m->rows * m->ld < (int64_t)0x7fffffff;
}

ScoreGrid score_grid(const GridSlot& g)
{
    ScoreGrid s;
    s.edt = g.d_edt;
    s.rows = g.meta.rows;
    s.cols = g.meta.cols;
    s.ld = g.meta.ld;
    s.ipix = 1 / g.meta.pixel;   // main.c:383 — one float division on the host
    s.min_x = g.meta.min_x;
    s.min_y = g.meta.min_y;
    return s;
}

int check_score_inputs(slam_engine* e, int slot)
{
    if (!slot_ok(slot)) return SLAM_ERR_INVALID_ARG;
    if (!e->grid[slot].ready || e->nbeams < 0) return SLAM_ERR_NOT_READY;
    return SLAM_OK;
}

// The grid as the scorers of MANY poses want it: with the byte-per-cell copy (kernels.h: ScoreGrid::packed) when the grid
// has one.  The copy is made on the first such call after the grid changed — two small launches and ONE wait for their verdict
// (does the 256-entry table give every cell back bit for bit?), then nothing until the grid changes again.  Few poses (the
// one-wavefront-per-pose kernel, the lattice) keep the float grid.  SLAM_SCORE_PACKED=0: never (measurements).
int many_pose_grid(slam_engine* e, int slot, int nposes, ScoreGrid* out)
{
    GridSlot& g = e->grid[slot];
    *out = score_grid(g);
    static const bool enabled = !(getenv("SLAM_SCORE_PACKED") && atoi(getenv("SLAM_SCORE_PACKED")) == 0);
    static const int wave_max = getenv("SLAM_SCORE_WAVE_MAX") ? atoi(getenv("SLAM_SCORE_WAVE_MAX")) : 3072;   // score_body.h: kWaveMaxPoses
    if (!enabled || nposes < wave_max || g.meta.rows < 8 || g.meta.cols < 16) return SLAM_OK;
    const int strip_bytes = 16 * ((g.meta.rows + 7) / 8 * 8);
    if (strip_bytes >= (1 << 24)) return SLAM_OK;   // 24-bit multiply in the scorer's cell offset
    if (g.packed_state == 0) {
        const size_t bytes = edt_packed_bytes(g.meta.rows, g.meta.cols);
        if (g.packed_buf.cap < bytes || g.table_buf.cap < 1024 + 8) {
            HIP_TRY(hipStreamSynchronize(e->stream));   // an earlier launch may still read the old copy
            HIP_TRY(g.packed_buf.ensure(bytes));
            HIP_TRY(g.table_buf.ensure(1024 + 8));
        }
        uint32_t* flag = reinterpret_cast<uint32_t*>(g.table_buf.as<float>() + 256);
        HIP_TRY(launch_edt_pack(e->stream, g.d_edt, g.meta.ld, g.meta.rows, g.meta.cols, g.packed_buf.as<uint8_t>(),
                                g.table_buf.as<float>(), flag));
        uint32_t verdict[2] = { 0, 1 };
        HIP_TRY(hipMemcpyAsync(verdict, flag, sizeof verdict, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        g.packed_state = verdict[1] == 0 ? 1 : 2;
    }
    if (g.packed_state == 1) {
        out->packed = g.packed_buf.as<uint8_t>();
        out->table = g.table_buf.as<float>();
        out->strip_bytes = strip_bytes;
    }
    return SLAM_OK;
}

// host-side Philox4x32-10 for the comb offset
void philox_host(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

}  // namespace

// Every pinned host buffer of the engine comes out of ONE allocation (mapped into the device): an allocation of pinned host
// memory is answered by the driver some 10-50 ms later with a 65-80 ms hold of the process's queues (DESIGN.md section 8,
// profiles/r03_stall_trigger.txt), so the engine makes one, when it is created, instead of eight.
static hipError_t engine_host_block(slam_engine* e)
{
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_fm = up(sizeof(float) * (kFmIn + kFmOut + 4 + kFmPair)), b_plan = up(sizeof(int32_t) * (SLAM_PLAN_WORDS(kMaxRanks) + 1)),
                 b_small = 256, b_res = 256, b_stage = up(sizeof(float) * kStageSlots * kStageFloats);
    const size_t total = b_fm + b_plan + 3 * b_small + b_res + b_stage;
    hipError_t err = hipHostMalloc(&e->h_block, total, hipHostMallocMapped);
    if (err != hipSuccess) return err;
    void* dblock = nullptr;
    if ((err = hipHostGetDevicePointer(&dblock, e->h_block, 0)) != hipSuccess) return err;
    char *h = static_cast<char*>(e->h_block), *d = static_cast<char*>(dblock);
    size_t off = 0;
    auto take = [&](size_t bytes, auto*& hp, auto*& dp) {
        hp = reinterpret_cast<std::remove_reference_t<decltype(hp)>>(h + off);
        dp = reinterpret_cast<std::remove_reference_t<decltype(dp)>>(d + off);
        off += bytes;
    };
    take(b_fm, e->h_fm, e->d_hfm);
    take(b_plan, e->h_plan, e->d_hplan);
    take(b_small, e->h_gate, e->d_hgate);
    take(b_small, e->h_heads, e->d_hheads);
    take(b_small, e->h_obs, e->d_hobs);
    take(b_res, e->h_pf_res, e->d_hpf_res);
    e->h_stage = reinterpret_cast<float*>(h + off);
    memset(e->h_block, 0, total);
    return hipSuccess;
}

extern "C" {

int slam_abi_version(void) { return SLAM_ABI_VERSION; }

const char* slam_status_string(int status)
{
    switch (status) {
    case SLAM_OK: return "ok";
    case SLAM_ERR_NO_DEVICE: return "no usable gfx950 device";
    case SLAM_ERR_INVALID_ARG: return "invalid argument";
    case SLAM_ERR_HIP: return "HIP runtime error";
    case SLAM_ERR_NOT_READY: return "stage inputs not provided yet";
    case SLAM_ERR_CAPACITY: return "capacity exceeded";
    case SLAM_ERR_COMM: return "exchange between ranks failed";
    default: return "unknown status";
    }
}

// why the last slam_engine_create of this thread failed (there is no engine to ask then): slam_last_error(NULL)
static thread_local char g_create_err[256] = "";

const char* slam_last_error(const slam_engine* e) { return e ? e->err : g_create_err; }

int slam_engine_create(int device, slam_engine** out)
{
    if (!out) return SLAM_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    g_create_err[0] = 0;
    hipError_t herr = hipGetDeviceCount(&ndev);
    if (herr != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        snprintf(g_create_err, sizeof g_create_err, "device %d of %d (hipGetDeviceCount: %s)", device, ndev, hipGetErrorString(herr));
        (void)hipGetLastError();
        return SLAM_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if ((herr = hipSetDevice(device)) != hipSuccess || (herr = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
        snprintf(g_create_err, sizeof g_create_err, "hipSetDevice / hipGetDeviceProperties(%d): %s", device, hipGetErrorString(herr));
        (void)hipGetLastError();
        return SLAM_ERR_NO_DEVICE;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {   // code objects are gfx950-only
        snprintf(g_create_err, sizeof g_create_err, "device %d is %s, not gfx950", device, prop.gcnArchName);
        return SLAM_ERR_NO_DEVICE;
    }
    slam_engine* e = new (std::nothrow) slam_engine();
    if (!e) return SLAM_ERR_HIP;
    e->device = device;
    if (hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess ||
        e->fm_buf.ensure(sizeof(float) * (kFmIn + kFmOut)) != hipSuccess ||
        e->fm_work.ensure(sizeof(float) * 2 * kLattice * SLAM_MAX_BEAMS) != hipSuccess ||   // (the hits of two sweeps: the chained pair)
        engine_host_block(e) != hipSuccess ||
        e->gate_buf.ensure(kGateBufWords * sizeof(int32_t)) != hipSuccess ||   // flag | ticket | accumulators: see kernels.h
        e->heads_buf.ensure(2 * sizeof(int32_t)) != hipSuccess ||
        hipMemset(e->heads_buf.p, 0, 2 * sizeof(int32_t)) != hipSuccess ||   // the gate's flag + the ticket word of quantise_scan_kernel
        e->scan_buf.ensure(sizeof(float) * 2 * SLAM_MAX_BEAMS) != hipSuccess) {
        snprintf(g_create_err, sizeof g_create_err, "allocating the engine's buffers on device %d: %s", device,
                 hipGetErrorString(hipGetLastError()));
        slam_engine_destroy(e);
        return SLAM_ERR_NO_DEVICE;
    }
    memset(e->h_fm, 0, sizeof(float) * (kFmIn + kFmOut + 4 + kFmPair));   // arrival flag starts at 0, sequence numbers at 1
    memset(e->h_plan, 0, sizeof(int32_t) * (SLAM_PLAN_WORDS(kMaxRanks) + 1));
    e->h_gate[0] = 1;
    e->h_gate[1] = 0;
    e->h_heads[0] = 0;
    e->h_heads[1] = -1;   // nothing known yet
    e->h_obs[0] = 0;
    e->h_obs[1] = -1;
    if (getenv("SLAM_EKF_INPLACE")) e->ekf_inplace_form = atoi(getenv("SLAM_EKF_INPLACE"));
    if (getenv("SLAM_PF_PAGED")) e->pf_paged = atoi(getenv("SLAM_PF_PAGED")) != 0;
    {
        const int32_t one[kGateBufWords] = { 1 };   // "the previous frame resampled": nothing is carried into the first frame; the rest 0
        if (hipMemcpy(e->gate_buf.p, one, sizeof one, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            slam_engine_destroy(e);
            return SLAM_ERR_NO_DEVICE;
        }
    }
    e->stream = e->own_stream;
    if (getenv("SLAM_EKF_GROUP")) e->ekf_form = atoi(getenv("SLAM_EKF_GROUP"));
    *out = e;
    return SLAM_OK;
}

int slam_engine_destroy(slam_engine* e)
{
    if (!e) return SLAM_OK;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    for (auto& g : e->grid) {
        g.occ_buf.release();
        g.edt_buf.release();
    }
    e->scan_buf.release();
    e->obs_buf.release();
    e->fm_buf.release();
    e->fm_work.release();
    e->scratch.release();
    e->bmax_buf.release();
    e->scan_state.release();
    e->ll_buf.release();
    e->shard_buf.release();
    e->first_buf.release();
    for (auto& ev : e->stage_ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& b : e->host_io) b.release();
    for (auto& pool : e->prof_pool)
        for (auto& p : pool) {
            (void)hipEventDestroy(p.start);
            (void)hipEventDestroy(p.stop);
        }
    if (e->h_block) (void)hipHostFree(e->h_block);   // every pinned buffer of the engine at once
    e->obs_list.release();
    e->heads_buf.release();
    e->gate_buf.release();
    e->carry_buf.release();
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
    return SLAM_OK;
}

int slam_profile_enable(slam_engine* e, int mask)
{
    ENTER(e);
    e->prof_mask = mask;
    return SLAM_OK;
}

int slam_profile_read(slam_engine* e, int kernel, double* total_ms, int64_t* launches)
{
    ENTER(e);
    if (kernel < 0 || kernel >= SLAM_PROF_COUNT || !total_ms || !launches) return SLAM_ERR_INVALID_ARG;
    HIP_TRY(hipStreamSynchronize(e->stream));
    double sum = 0.0;
    for (size_t i = 0; i < e->prof_used[kernel]; ++i) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, e->prof_pool[kernel][i].start, e->prof_pool[kernel][i].stop));
        sum += (double)ms;
    }
    *total_ms = sum;
    *launches = (int64_t)e->prof_used[kernel];
    e->prof_used[kernel] = 0;
    return SLAM_OK;
}

int slam_profile_bracket_overhead(slam_engine* e, double* overhead_ms)
{
    ENTER(e);
    if (!overhead_ms) return SLAM_ERR_INVALID_ARG;
    enum { kPairs = 64 };
    hipEvent_t ev[2 * kPairs];
    for (auto& x : ev) HIP_TRY(hipEventCreate(&x));
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int k = 0; k < kPairs; ++k) {
        HIP_TRY(hipEventRecord(ev[2 * k], e->stream));
        HIP_TRY(hipEventRecord(ev[2 * k + 1], e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    double sum = 0.0;
    for (int k = 0; k < kPairs; ++k) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]));
        sum += (double)ms;
    }
    for (auto& x : ev) (void)hipEventDestroy(x);
    *overhead_ms = sum / kPairs;
    return SLAM_OK;
}

int slam_profile_copy_ceiling(slam_engine* e, const float* d_src, float* d_dst, int64_t rows, int plane_stride, int reps,
                              double* ms_per_copy)
{
    ENTER(e);
    if (!d_src || !d_dst || d_src == d_dst || rows <= 0 || rows > 0x7fffffff || plane_stride < 128 || plane_stride % 128 ||
        reps <= 0 || !ms_per_copy)
        return SLAM_ERR_INVALID_ARG;
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    int rc = SLAM_OK;
    auto ok = [&](hipError_t err, const char* what) {
        if (err != hipSuccess && rc == SLAM_OK) rc = fail_hip(e, err, what);
        return err == hipSuccess;
    };
    for (int r = 0; r < 2 && rc == SLAM_OK; ++r) ok(launch_copy_rows(e->stream, d_src, d_dst, (int)rows, plane_stride), "copy_rows");
    ok(hipEventRecord(a, e->stream), "hipEventRecord");
    for (int r = 0; r < reps && rc == SLAM_OK; ++r) ok(launch_copy_rows(e->stream, d_src, d_dst, (int)rows, plane_stride), "copy_rows");
    ok(hipEventRecord(b, e->stream), "hipEventRecord");
    ok(hipEventSynchronize(b), "hipEventSynchronize");
    float ms = 0.0f;
    if (rc == SLAM_OK) ok(hipEventElapsedTime(&ms, a, b), "hipEventElapsedTime");
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms_per_copy = (double)ms / reps;
    return rc;
}

int slam_selftest_reciprocal(slam_engine* e, int64_t* mismatches, int64_t* checked)
{
    ENTER(e);
    if (!mismatches || !checked) return SLAM_ERR_INVALID_ARG;
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 16));
    int rc = SLAM_OK;
    unsigned long long h[2] = { 0, 0 };
    hipError_t err = hipMemsetAsync(d, 0, 16, e->stream);
    if (err == hipSuccess) err = launch_selftest_reciprocal(e->stream, d);
    if (err == hipSuccess) err = hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    if (err != hipSuccess) rc = fail_hip(e, err, "slam_selftest_reciprocal");
    (void)hipFree(d);
    *mismatches = (int64_t)h[0];
    *checked = (int64_t)h[1];
    return rc;
}

int slam_engine_set_stream(slam_engine* e, void* hip_stream)
{
    ENTER(e);
    e->stream = hip_stream == SLAM_OWN_STREAM ? e->own_stream : static_cast<hipStream_t>(hip_stream);
    return SLAM_OK;
}

int slam_engine_sync(slam_engine* e)
{
    ENTER(e);
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SLAM_OK;
}

/* ------------------------------------------------------------------ EDT */

int slam_edt_dev(slam_engine* e, const int32_t* d_occ, int ld, int rows, int cols, float cap, float* d_out)
{
    ENTER(e);
    if (!d_occ || !d_out || rows < 0 || cols < 0 || ld < cols || !(cap >= 0.0f)) return SLAM_ERR_INVALID_ARG;
    if (ceilf(cap) > (float)EDT_MAX_RADIUS) return SLAM_ERR_CAPACITY;
    HIP_TRY(launch_edt(e->stream, d_occ, ld, rows, cols, cap, d_out, e->prof_next(SLAM_PROF_EDT)));
    for (GridSlot& g : e->grid)   // an adopted grid rebuilt in place: its packed copy is stale
        if (g.ready && g.d_edt == d_out) g.packed_state = 0;
    return SLAM_OK;
}

int slam_edt_host(slam_engine* e, const int32_t* occ, int ld, int rows, int cols, float cap, float* out)
{
    ENTER(e);
    if (!occ || !out || rows < 0 || cols < 0 || ld < cols || !(cap >= 0.0f)) return SLAM_ERR_INVALID_ARG;
    if (ceilf(cap) > (float)EDT_MAX_RADIUS) return SLAM_ERR_CAPACITY;
    if (rows == 0 || cols == 0) return SLAM_OK;
    const size_t cells = (size_t)rows * ld;
    HIP_TRY(e->host_io[0].ensure(cells * sizeof(int32_t)));
    HIP_TRY(e->host_io[1].ensure(cells * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(e->host_io[0].p, occ, cells * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(launch_edt(e->stream, e->host_io[0].as<int32_t>(), ld, rows, cols, cap, e->host_io[1].as<float>()));
    // only the rows x cols rectangle belongs to the caller's output (cells outside keep their content, Q7)
    HIP_TRY(hipMemcpy2DAsync(out, (size_t)ld * sizeof(float), e->host_io[1].p, (size_t)ld * sizeof(float),
                             (size_t)cols * sizeof(float), (size_t)rows, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SLAM_OK;
}

/* ------------------------------------------------------------------ grids + scan */

int slam_grid_upload_host(slam_engine* e, int slot, const int32_t* occ, const slam_grid_meta* meta, float cap,
                          float* edt_out)
{
    ENTER(e);
    if (!slot_ok(slot) || !occ || !meta_ok(meta) || !(cap >= 0.0f)) return SLAM_ERR_INVALID_ARG;
    if (ceilf(cap) > (float)EDT_MAX_RADIUS) return SLAM_ERR_CAPACITY;
    GridSlot& g = e->grid[slot];
    const size_t cells = (size_t)(meta->rows > 0 ? meta->rows : 1) * meta->ld;
    HIP_TRY(g.occ_buf.ensure(cells * sizeof(int32_t)));
    HIP_TRY(g.edt_buf.ensure(cells * sizeof(float)));
    if (meta->rows > 0) {
        HIP_TRY(hipMemcpyAsync(g.occ_buf.p, occ, (size_t)meta->rows * meta->ld * sizeof(int32_t), hipMemcpyHostToDevice,
                               e->stream));
        HIP_TRY(launch_edt(e->stream, g.occ_buf.as<int32_t>(), meta->ld, meta->rows, meta->cols, cap,
                           g.edt_buf.as<float>()));
    }
    g.meta = *meta;
    g.d_edt = g.edt_buf.as<float>();
    g.ready = true;
    g.packed_state = 0;
    if (edt_out && meta->rows > 0 && meta->cols > 0) {
        HIP_TRY(hipMemcpy2DAsync(edt_out, (size_t)meta->ld * sizeof(float), g.edt_buf.p,
                                 (size_t)meta->ld * sizeof(float), (size_t)meta->cols * sizeof(float),
                                 (size_t)meta->rows, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return SLAM_OK;
}

int slam_grid_set_dev(slam_engine* e, int slot, const float* d_edt, const slam_grid_meta* meta)
{
    ENTER(e);
    if (!slot_ok(slot) || !d_edt || !meta_ok(meta)) return SLAM_ERR_INVALID_ARG;
    GridSlot& g = e->grid[slot];
    g.meta = *meta;
    g.d_edt = d_edt;
    g.ready = true;
    g.packed_state = 0;   // the scorers' packed copy is made from the new contents on their next call
    return SLAM_OK;
}

int slam_grid_set_meta(slam_engine* e, int slot, const slam_grid_meta* meta)
{
    ENTER(e);
    if (!slot_ok(slot) || !meta_ok(meta)) return SLAM_ERR_INVALID_ARG;
    GridSlot& g = e->grid[slot];
    if (!g.ready) return SLAM_ERR_NOT_READY;
    if (meta->rows != g.meta.rows || meta->cols != g.meta.cols || meta->ld != g.meta.ld) return SLAM_ERR_INVALID_ARG;
    g.meta = *meta;
    return SLAM_OK;
}

int slam_scan_upload_host(slam_engine* e, const float* bx, const float* by, int nbeams)
{
    ENTER(e);
    if (nbeams < 0 || (nbeams > 0 && (!bx || !by))) return SLAM_ERR_INVALID_ARG;
    if (nbeams > SLAM_MAX_BEAMS) return SLAM_ERR_CAPACITY;
    float* d = e->scan_buf.as<float>();
    if (nbeams > 0) {
        // pinned staging -> ONE host-to-device copy per frame (bx | by back to back)
        float* h = e->stage_acquire();
        memcpy(h, bx, sizeof(float) * nbeams);
        memcpy(h + nbeams, by, sizeof(float) * nbeams);
        HIP_TRY(hipMemcpyAsync(d, h, sizeof(float) * 2 * nbeams, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(e->stage_release(h));
    }
    e->d_bx = d;
    e->d_by = d + nbeams;
    e->nbeams = nbeams;
    return SLAM_OK;
}

int slam_scan_set_dev(slam_engine* e, const float* d_bx, const float* d_by, int nbeams)
{
    ENTER(e);
    if (nbeams < 0 || (nbeams > 0 && (!d_bx || !d_by))) return SLAM_ERR_INVALID_ARG;
    if (nbeams > SLAM_MAX_BEAMS) return SLAM_ERR_CAPACITY;
    e->d_bx = d_bx;
    e->d_by = d_by;
    e->nbeams = nbeams;
    return SLAM_OK;
}

/* ------------------------------------------------------------------ score */

int slam_score_poses_cs_dev(slam_engine* e, int slot, const float* d_x, const float* d_y, const float* d_ct,
                            const float* d_st, int nposes, float* d_score, int32_t* d_count)
{
    ENTER(e);
    if (nposes < 0 || (nposes > 0 && (!d_x || !d_y || !d_ct || !d_st || !d_score || !d_count)))
        return SLAM_ERR_INVALID_ARG;
    if (int rc = check_score_inputs(e, slot)) return rc;
    ScoreGrid sg;
    if (int rc = many_pose_grid(e, slot, nposes, &sg)) return rc;
    HIP_TRY(launch_score_poses(e->stream, sg, e->d_bx, e->d_by, e->nbeams, d_x, d_y, d_ct, d_st,
                               nposes, d_score, d_count, e->prof_next(SLAM_PROF_SCORE)));
    return SLAM_OK;
}

int slam_score_poses_dev(slam_engine* e, int slot, const float* d_x, const float* d_y, const float* d_theta,
                         int nposes, float* d_score, int32_t* d_count)
{
    ENTER(e);
    if (nposes < 0 || (nposes > 0 && (!d_x || !d_y || !d_theta || !d_score || !d_count))) return SLAM_ERR_INVALID_ARG;
    if (int rc = check_score_inputs(e, slot)) return rc;
    ScoreGrid sg;
    if (int rc = many_pose_grid(e, slot, nposes, &sg)) return rc;
    HIP_TRY(launch_score_poses(e->stream, sg, e->d_bx, e->d_by, e->nbeams, d_x, d_y, d_theta,
                               nullptr, nposes, d_score, d_count, e->prof_next(SLAM_PROF_SCORE)));
    return SLAM_OK;
}

static int score_host_common(slam_engine* e, int slot, const float* x, const float* y, const float* a,
                             const float* b, int nposes, float* score, int32_t* count)
{
    if (nposes < 0 || (nposes > 0 && (!x || !y || !a || !score || !count))) return SLAM_ERR_INVALID_ARG;
    if (int rc = check_score_inputs(e, slot)) return rc;
    if (nposes == 0) return SLAM_OK;
    const size_t bytes = sizeof(float) * (size_t)nposes;
    const float* src[4] = { x, y, a, b };
    for (int k = 0; k < 6; ++k) HIP_TRY(e->host_io[k].ensure(bytes));
    for (int k = 0; k < 4; ++k)
        if (src[k]) HIP_TRY(hipMemcpyAsync(e->host_io[k].p, src[k], bytes, hipMemcpyHostToDevice, e->stream));
    ScoreGrid sg;
    if (int rc = many_pose_grid(e, slot, nposes, &sg)) return rc;
    HIP_TRY(launch_score_poses(e->stream, sg, e->d_bx, e->d_by, e->nbeams,
                               e->host_io[0].as<float>(), e->host_io[1].as<float>(), e->host_io[2].as<float>(),
                               b ? e->host_io[3].as<float>() : nullptr, nposes, e->host_io[4].as<float>(),
                               e->host_io[5].as<int32_t>()));
    HIP_TRY(hipMemcpyAsync(score, e->host_io[4].p, bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(count, e->host_io[5].p, bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SLAM_OK;
}

int slam_score_poses_cs_host(slam_engine* e, int slot, const float* x, const float* y, const float* ct,
                             const float* st, int nposes, float* score, int32_t* count)
{
    ENTER(e);
    if (nposes > 0 && !st) return SLAM_ERR_INVALID_ARG;
    return score_host_common(e, slot, x, y, ct, st, nposes, score, count);
}

int slam_score_poses_host(slam_engine* e, int slot, const float* x, const float* y, const float* theta, int nposes,
                          float* score, int32_t* count)
{
    ENTER(e);
    return score_host_common(e, slot, x, y, theta, nullptr, nposes, score, count);
}

int slam_pose_hits_host(slam_engine* e, int slot, float x, float y, float ct, float st, float* hits, int32_t* count)
{
    ENTER(e);
    if (!hits || !count) return SLAM_ERR_INVALID_ARG;
    if (int rc = check_score_inputs(e, slot)) return rc;
    float* d_in = e->fm_buf.as<float>();
    float* d_out = d_in + kFmIn;
    float* h_in = e->h_fm;
    float* h_out = e->h_fm + kFmIn;
    h_in[4 * kLattice + 0] = x;
    h_in[4 * kLattice + 1] = y;
    h_in[4 * kLattice + 2] = ct;
    h_in[4 * kLattice + 3] = st;
    HIP_TRY(hipMemcpyAsync(d_in + 4 * kLattice, h_in + 4 * kLattice, 4 * sizeof(float), hipMemcpyHostToDevice,
                           e->stream));
    HIP_TRY(launch_pose_hits(e->stream, score_grid(e->grid[slot]), e->d_bx, e->d_by, e->nbeams, d_in + 4 * kLattice,
                             d_out + 2 * kLattice + 1, reinterpret_cast<int32_t*>(d_out + 2 * kLattice)));
    HIP_TRY(hipMemcpyAsync(h_out + 2 * kLattice, d_out + 2 * kLattice, sizeof(float) * (1 + (size_t)e->nbeams),
                           hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    memcpy(count, h_out + 2 * kLattice, sizeof(int32_t));
    if (*count > 0) memcpy(hits, h_out + 2 * kLattice + 1, sizeof(float) * (size_t)*count);
    return SLAM_OK;
}

}  // extern "C"

// ---- internal entry points shared with mapper.hip (C++ linkage, declared in engine_internal.h)
int slam_engine_fail_hip(slam_engine* e, hipError_t err, const char* what) { return fail_hip(e, err, what); }

slam::ScoreGrid slam_engine_score_grid(const slam_engine* e, int slot) { return score_grid(e->grid[slot]); }

int slam_engine_fastmatch(slam_engine* e, int slot, const float* d_bx, const float* d_by, int nbeams_max,
                          const int32_t* d_nbeams, const float pose[3], const float res[3], float out_pose[3],
                          float* best_hits, int32_t* best_hits_size, float* best_score, float* d_hits_persist)
{
    // main.c:386-387, :424-426 — the lattice is laid out once around the input pose; res[0] steps x
    // AND y, res[2] steps theta, res[1] is never read.  Heading trig with the host libm, as the
    // reference does (main.c:433-435).
    const float t = res[0], r = res[2];
    const float th[3] = { pose[2] - r, pose[2], pose[2] + r };
    const float xs[3] = { pose[0] - t, pose[0], pose[0] + t };
    const float ys[3] = { pose[1] - t, pose[1], pose[1] + t };
    float* h_in = e->h_fm;
    float* h_out = e->h_fm + kFmIn;
    for (int a = 0; a < 3; ++a) {
        const float c = cosf(th[a]), s = sinf(th[a]);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const int k = (a * 3 + i) * 3 + j;   // evaluation order theta, x, y (main.c:443-487)
                h_in[k] = xs[i];
                h_in[kLattice + k] = ys[j];
                h_in[2 * kLattice + k] = c;
                h_in[3 * kLattice + k] = s;
            }
    }
    float* d_out = e->fm_buf.as<float>() + kFmIn;
    const ScoreGrid g = score_grid(e->grid[slot]);
    // zero-copy I/O: the kernels read the 27 candidates from, and deliver their result to, pinned host memory
    // mapped into the device; the host waits for the arrival flag instead of a copy + stream synchronisation
    volatile uint32_t* h_flag = reinterpret_cast<volatile uint32_t*>(e->h_fm + kFmIn + kFmOut);
    const uint32_t seq = ++e->fm_seq;
    HIP_TRY(launch_lattice(e->stream, g, d_bx, d_by, nbeams_max, d_nbeams, e->d_hfm, e->fm_work.as<float>(), d_out,
                           d_hits_persist, e->d_hfm + kFmIn, reinterpret_cast<uint32_t*>(e->d_hfm + kFmIn + kFmOut), seq));
    {
        bool arrived = false;
        for (long spin = 0; spin < 400000000L; ++spin) {   // bounded: a few seconds at most
            if (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) == seq) { arrived = true; break; }
        }
        if (!arrived) {   // the launch failed or the device is wedged: let the runtime tell us
            HIP_TRY(hipStreamSynchronize(e->stream));
            if (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != seq) return fail_hip(e, hipErrorUnknown, "lattice result flag");
        }
    }

    // main.c:549-563 — strict '<' keeps the first of equal scores
    float best = INFINITY;
    int best_k = -1;
    for (int k = 0; k < kLattice; ++k)
        if (h_out[k] < best) {
            best = h_out[k];
            best_k = k;
        }
    if (best_k >= 0) {
        out_pose[0] = h_in[best_k];
        out_pose[1] = h_in[kLattice + best_k];
        out_pose[2] = th[best_k / 9];
        memcpy(best_hits_size, h_out + kLattice + best_k, sizeof(int32_t));
    } else {   // nothing beat +inf (NaN scores): the reference returns the input pose, size untouched
        out_pose[0] = pose[0];
        out_pose[1] = pose[1];
        out_pose[2] = pose[2];
    }
    // the caller's hit buffer ends up exactly as the reference's shared scratch does: the
    // prefix every candidate overwrote, last writer wins; entries beyond the longest candidate untouched
    int32_t maxc;
    memcpy(&maxc, h_out + 2 * kLattice, sizeof maxc);
    if (maxc > 0 && best_hits) memcpy(best_hits, h_out + 2 * kLattice + 1, sizeof(float) * (size_t)maxc);
    if (best_score) *best_score = best;
    return SLAM_OK;
}

// main.c:901-924 as ONE round trip: FastMatch(pose, res1) on grid slot1, then FastMatch2(its result, res2) on grid slot2 — the
// second call's lattice is laid out on the device around the first call's best candidate, so the host waits once instead of
// twice (a call is bound by that wait: 30 us, of which the kernels are a third).  What the reference computes with libm
// stays on the host: the first call's best heading is one of three values, so the second call's three headings are among
// NINE known before the launch; the host sends the cosines and sines of all nine and the device picks its three.  The
// candidates' x and y are one float add / subtract each, the same on the device.  The host repeats both arg-min decisions
// on the scores it receives (strict '<', the first of equals; nothing below +inf: the input pose and the previous size) —
// the device's choice of the first call's winner is the same computation on the same floats.
int slam_engine_fastmatch_pair(slam_engine* e, int slot1, int slot2, const float* d_bx, const float* d_by, int nbeams_max,
                               const int32_t* d_nbeams, const float pose[3], const float res1[3], const float res2[3],
                               float out_pose[3], int32_t* best_hits_size, float* d_hits_persist)
{
    const float t1 = res1[0], r1 = res1[2], t2 = res2[0], r2 = res2[2];
    const float th1[3] = { pose[2] - r1, pose[2], pose[2] + r1 };
    const float xs[3] = { pose[0] - t1, pose[0], pose[0] + t1 };
    const float ys[3] = { pose[1] - t1, pose[1], pose[1] + t1 };
    float* h_in = e->h_fm;
    float* h_out2 = e->h_fm + kFmIn;
    float* h_pair_in = e->h_fm + kFmIn + kFmOut + 4;
    float* h_out1 = h_pair_in + kFmPairIn;
    float th2[3][3];
    for (int a = 0; a < 3; ++a) {
        const float c = cosf(th1[a]), s = sinf(th1[a]);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const int k = (a * 3 + i) * 3 + j;   // evaluation order theta, x, y (main.c:443-487)
                h_in[k] = xs[i];
                h_in[kLattice + k] = ys[j];
                h_in[2 * kLattice + k] = c;
                h_in[3 * kLattice + k] = s;
            }
        th2[a][0] = th1[a] - r2;
        th2[a][1] = th1[a];
        th2[a][2] = th1[a] + r2;
        for (int b = 0; b < 3; ++b) {
            h_pair_in[a * 3 + b] = cosf(th2[a][b]);
            h_pair_in[9 + a * 3 + b] = sinf(th2[a][b]);
        }
    }
    h_pair_in[18] = t2;
    float* d_out1 = e->fm_buf.as<float>();           // the first call's scores | counts | maxcount (kFmIn floats are room enough)
    float* d_out2 = e->fm_buf.as<float>() + kFmIn;   // the second call's
    float* work1 = e->fm_work.as<float>();
    float* work2 = work1 + (size_t)kLattice * SLAM_MAX_BEAMS;
    volatile uint32_t* h_flag = reinterpret_cast<volatile uint32_t*>(e->h_fm + kFmIn + kFmOut);
    const uint32_t seq = ++e->fm_seq;
    HIP_TRY(launch_lattice_pair(e->stream, score_grid(e->grid[slot1]), score_grid(e->grid[slot2]), d_bx, d_by, nbeams_max, d_nbeams, e->d_hfm,
                                e->d_hfm + kFmIn + kFmOut + 4, work1, work2, d_out1, d_out2, d_hits_persist,
                                e->d_hfm + kFmIn + kFmOut + 4 + kFmPairIn, e->d_hfm + kFmIn,
                                reinterpret_cast<uint32_t*>(e->d_hfm + kFmIn + kFmOut), seq));
    {
        bool arrived = false;
        for (long spin = 0; spin < 400000000L; ++spin) {   // bounded: a few seconds at most
            if (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) == seq) { arrived = true; break; }
        }
        if (!arrived) {
            HIP_TRY(hipStreamSynchronize(e->stream));
            if (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != seq) return fail_hip(e, hipErrorUnknown, "lattice result flag");
        }
    }
    // the first call (main.c:549-563): strict '<' keeps the first of equal scores; nothing below +inf: the input pose, size untouched
    float best = INFINITY;
    int k1 = -1;
    for (int k = 0; k < kLattice; ++k)
        if (h_out1[k] < best) {
            best = h_out1[k];
            k1 = k;
        }
    float p1[3] = { pose[0], pose[1], pose[2] };
    int a1 = 1;
    if (k1 >= 0) {
        p1[0] = h_in[k1];
        p1[1] = h_in[kLattice + k1];
        a1 = k1 / 9;
        p1[2] = th1[a1];
        memcpy(best_hits_size, h_out1 + kLattice + k1, sizeof(int32_t));
    }
    // the second call, laid out around p1 (the device built the same table around the same candidate)
    const float xs2[3] = { p1[0] - t2, p1[0], p1[0] + t2 };
    const float ys2[3] = { p1[1] - t2, p1[1], p1[1] + t2 };
    best = INFINITY;
    int k2 = -1;
    for (int k = 0; k < kLattice; ++k)
        if (h_out2[k] < best) {
            best = h_out2[k];
            k2 = k;
        }
    if (k2 >= 0) {
        out_pose[0] = xs2[(k2 / 3) % 3];
        out_pose[1] = ys2[k2 % 3];
        out_pose[2] = th2[a1][k2 / 9];
        memcpy(best_hits_size, h_out2 + kLattice + k2, sizeof(int32_t));
    } else {
        out_pose[0] = p1[0];
        out_pose[1] = p1[1];
        out_pose[2] = p1[2];
    }
    return SLAM_OK;
}

extern "C" {

int slam_fastmatch_host(slam_engine* e, int slot, const float pose[3], const float res[3], float out_pose[3],
                        float* best_hits, int32_t* best_hits_size, float* best_score)
{
    ENTER(e);
    if (!pose || !res || !out_pose || !best_hits || !best_hits_size) return SLAM_ERR_INVALID_ARG;
    if (int rc = check_score_inputs(e, slot)) return rc;
    return slam_engine_fastmatch(e, slot, e->d_bx, e->d_by, e->nbeams, nullptr, pose, res, out_pose, best_hits,
                                 best_hits_size, best_score, nullptr);
}

/* ------------------------------------------------------------------ particle-filter stages */

int slam_motion_sample_dev(slam_engine* e, const float* d_src_x, const float* d_src_y, const float* d_src_th,
                           const int32_t* d_anc, float* d_x, float* d_y, float* d_th, int n, int64_t first_id,
                           const float dp[3], const float sigma[3], uint64_t seed, uint32_t frame)
{
    ENTER(e);
    if (n < 0 || first_id < 0 || !dp || !sigma ||
        (n > 0 && (!d_src_x || !d_src_y || !d_src_th || !d_x || !d_y || !d_th)))
        return SLAM_ERR_INVALID_ARG;
    if (d_anc && (d_src_x == d_x || d_src_y == d_y || d_src_th == d_th)) return SLAM_ERR_INVALID_ARG;   // gather in place
    HIP_TRY(launch_motion_sample(e->stream, d_src_x, d_src_y, d_src_th, d_anc, d_x, d_y, d_th, n, first_id, dp, sigma,
                                 seed, frame));
    return SLAM_OK;
}

int slam_motion_score_dev(slam_engine* e, int slot, const float* d_src_x, const float* d_src_y, const float* d_src_th,
                          const int32_t* d_anc, float* d_x, float* d_y, float* d_th, int n, int64_t first_id,
                          const float dp[3], const float sigma[3], uint64_t seed, uint32_t frame, float* d_score,
                          int32_t* d_count)
{
    return slam_motion_score_rider_dev(e, slot, d_src_x, d_src_y, d_src_th, d_anc, d_x, d_y, d_th, n, first_id, dp, sigma, seed, frame,
                                       d_score, d_count, nullptr, nullptr);
}

// ... with a paged session's free list in workgroups of the same launch (kernels.h: FreeListRider); *rode = false: the caller
// launches the list by itself
int slam_motion_score_rider_dev(slam_engine* e, int slot, const float* d_src_x, const float* d_src_y, const float* d_src_th,
                                const int32_t* d_anc, float* d_x, float* d_y, float* d_th, int n, int64_t first_id,
                                const float dp[3], const float sigma[3], uint64_t seed, uint32_t frame, float* d_score,
                                int32_t* d_count, const FreeListRider* rider, bool* rode)
{
    ENTER(e);
    if (rode) *rode = false;
    if (n < 0 || first_id < 0 || !dp || !sigma ||
        (n > 0 && (!d_src_x || !d_src_y || !d_src_th || !d_x || !d_y || !d_th || !d_score || !d_count)))
        return SLAM_ERR_INVALID_ARG;
    if (d_src_x == d_x || d_src_y == d_y || d_src_th == d_th) return SLAM_ERR_INVALID_ARG;   // several lanes re-read src
    if (int rc = check_score_inputs(e, slot)) return rc;
    MotionIO io{ d_src_x, d_src_y, d_src_th, d_anc, d_x, d_y, d_th, rider ? *rider : FreeListRider() };
    ScoreGrid sg;
    if (int rc = many_pose_grid(e, slot, n, &sg)) return rc;
    HIP_TRY(launch_motion_score(e->stream, sg, e->d_bx, e->d_by, e->nbeams, io, n, first_id, dp,
                                sigma, seed, frame, d_score, d_count, e->prof_next(SLAM_PROF_SCORE), rode));
    return SLAM_OK;
}

int slam_obs_upload_host(slam_engine* e, const int32_t* landmark_id, const float* zx, const float* zy, int nobs,
                         int nlandmarks)
{
    ENTER(e);
    if (nobs < 0 || nlandmarks < 0 || nobs > nlandmarks || (nobs > 0 && (!landmark_id || !zx || !zy)))
        return SLAM_ERR_INVALID_ARG;
    if (nlandmarks > SLAM_MAX_OBS) return SLAM_ERR_CAPACITY;
    // the engine works on a table indexed by landmark: zx[l], zy[l], NaN = no observation of l this frame
    float* h = e->stage_acquire();
    const size_t L = (size_t)nlandmarks;
    float* hx = h;
    float* hy = h + L;
    const float nan = std::numeric_limits<float>::quiet_NaN();
    for (size_t l = 0; l < L; ++l) hx[l] = hy[l] = nan;
    int rc = SLAM_OK;
    for (int k = 0; k < nobs && rc == SLAM_OK; ++k) {
        const int32_t id = landmark_id[k];
        if (id < 0 || id >= nlandmarks || hx[id] == hx[id] || zx[k] != zx[k] || zy[k] != zy[k])
            rc = SLAM_ERR_INVALID_ARG;   // out of range, listed twice, or a NaN measurement
        else {
            hx[id] = zx[k];
            hy[id] = zy[k];
        }
    }
    if (rc != SLAM_OK) {
        (void)e->stage_release(h);   // nothing was queued from this slot
        return rc;
    }
    if (e->obs_buf.cap < 2 * L * 4) {
        HIP_TRY(hipStreamSynchronize(e->stream));   // a running kernel may still read the old table
        HIP_TRY(e->obs_buf.ensure(2 * L * 4 > 8 ? 2 * L * 4 : 8));
    }
    if (L > 0) HIP_TRY(hipMemcpyAsync(e->obs_buf.as<float>(), h, sizeof(float) * 2 * L, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e->stage_release(h));
    e->d_obs_zx = e->obs_buf.as<float>();
    e->d_obs_zy = e->obs_buf.as<float>() + L;
    e->obs_nlandmarks = nlandmarks;
    e->obs_list_valid = false;
    e->obs_table_owned = true;
    return SLAM_OK;
}

int slam_obs_set_dev(slam_engine* e, const float* d_zx_by_landmark, const float* d_zy_by_landmark, int nlandmarks)
{
    ENTER(e);
    if (nlandmarks < 0 || (nlandmarks > 0 && (!d_zx_by_landmark || !d_zy_by_landmark))) return SLAM_ERR_INVALID_ARG;
    e->d_obs_zx = d_zx_by_landmark;
    e->d_obs_zy = d_zy_by_landmark;
    e->obs_nlandmarks = nlandmarks;
    e->obs_list_valid = false;
    e->obs_table_owned = false;   // the caller may rewrite the arrays between launches: a list made from them is never reused
    return SLAM_OK;
}

int slam_ekf_update_dev(slam_engine* e, const float* d_map_in, float* d_map_out, int64_t row_stride, int plane_stride,
                        int nlandmarks, const float* d_x, const float* d_y, const float* d_th, const int32_t* d_anc,
                        int n, float meas_var, float* d_loglik)
{
    ENTER(e);
    if (n < 0 || nlandmarks < 0 || plane_stride < nlandmarks || row_stride < 5 * (int64_t)plane_stride ||
        !(meas_var > 0.0f) || (n > 0 && (!d_map_in || !d_map_out || !d_x || !d_y || !d_th)))
        return SLAM_ERR_INVALID_ARG;
    if (d_anc && d_map_in == d_map_out) return SLAM_ERR_INVALID_ARG;
    if (e->obs_nlandmarks < 0 || e->obs_nlandmarks != nlandmarks) return SLAM_ERR_NOT_READY;
    if (n == 0) return SLAM_OK;
    HIP_TRY(e->ll_buf.ensure(sizeof(float) * (size_t)n));
    EkfArgs a;
    a.map_in = d_map_in;
    a.map_out = d_map_out;
    a.row_stride = row_stride;
    a.plane_stride = plane_stride;
    a.nlandmarks = nlandmarks;
    a.x = d_x;
    a.y = d_y;
    a.th = d_th;
    a.anc = d_anc;
    a.n = n;
    a.obs_zx = e->d_obs_zx;
    a.obs_zy = e->d_obs_zy;
    a.meas_var = meas_var;
    a.loglik = e->ll_buf.as<float>();   // what slam_logweight_ekf_dev will consume
    a.loglik_user = d_loglik;
    if (d_map_in == d_map_out) {
        // in place: whole rows, or — when the last list that was built had few observations — the observed landmarks only
        const bool can_list = nlandmarks <= kObsListMaxLandmarks;
        if (!e->obs_table_owned) e->obs_list_valid = false;   // the caller's arrays may have been rewritten since the last launch
        const bool sparse = can_list && (e->ekf_inplace_form >= 0 ? e->ekf_inplace_form == 1
                                                                  : e->h_obs[1] == nlandmarks && 4 * (int64_t)e->h_obs[0] <= nlandmarks);
        const bool build = can_list && !e->obs_list_valid && (sparse || e->ekf_inplace_form < 0);
        const size_t L = (size_t)nlandmarks;
        int32_t* li = nullptr;
        if (build || sparse) {
            if (e->obs_list.cap < 4 * (4 * L + 2)) {
                HIP_TRY(hipStreamSynchronize(e->stream));
                HIP_TRY(e->obs_list.ensure(4 * (4 * L + 2)));
                e->obs_list_valid = false;
            }
            li = e->obs_list.as<int32_t>();
        }
        auto build_list = [&]() -> hipError_t {
            e->obs_list_valid = true;
            return launch_build_obs_list(e->stream, e->d_obs_zx, e->d_obs_zy, nlandmarks, li, (float*)(li + L),
                                         (float*)(li + 2 * L), li + 3 * L, li + 4 * L, e->d_hobs);
        };
        if (sparse) {
            if (!e->obs_list_valid) HIP_TRY(build_list());
            HIP_TRY(launch_ekf_sparse(e->stream, a, li, (const float*)(li + L), (const float*)(li + 2 * L), li + 3 * L,
                                      li + 4 * L, e->prof_next(SLAM_PROF_EKF)));
        } else {
            HIP_TRY(launch_ekf_update(e->stream, a, e->prof_next(SLAM_PROF_EKF), 0));
            if (build) HIP_TRY(build_list());   // after the update: only the count for the next frames is wanted
        }
        e->ekf_inplace_launches[sparse ? 1 : 0]++;
    } else {
        const int group = nlandmarks > 128 ? e->ekf_group_size(n, d_anc != nullptr, plane_stride, false) : 0;
        HIP_TRY(launch_ekf_update(e->stream, a, e->prof_next(SLAM_PROF_EKF), group));
        e->ekf_form_launches[group ? 1 : 0]++;
    }
    e->ll_n = n;
    return SLAM_OK;
}

int slam_frame_front_dev(slam_engine* e, int slot, const float* d_src_x, const float* d_src_y, const float* d_src_th,
                         const int32_t* d_anc, float* d_x, float* d_y, float* d_th, int n, int64_t first_id, const float dp[3],
                         const float sigma[3], uint64_t seed, uint32_t frame, float* d_score, int32_t* d_count,
                         const float* d_map_in, float* d_map_out, int64_t row_stride, int plane_stride, int nlandmarks,
                         float meas_var, bool* launched, const slam::SplitIO* split)
{
    ENTER(e);
    *launched = false;
    static const int env_fusion = getenv("SLAM_FRAME_FUSION") ? atoi(getenv("SLAM_FRAME_FUSION")) : -1;
    if (!(env_fusion >= 0 ? env_fusion != 0 : e->frame_fusion)) return SLAM_OK;
    if (e->prof_mask & (1 << SLAM_PROF_SCORE)) return SLAM_OK;   // the score stage is being timed: it stays a launch of its own
    // the checks of slam_motion_score_dev and of slam_ekf_update_dev (out of place)
    if (n <= 0 || first_id < 0 || !dp || !sigma || !d_src_x || !d_src_y || !d_src_th || !d_x || !d_y || !d_th || !d_score ||
        !d_count || !d_anc || !d_map_in || !d_map_out || d_map_in == d_map_out)
        return SLAM_OK;   // the two calls will say what is wrong
    if (d_src_x == d_x || d_src_y == d_y || d_src_th == d_th) return SLAM_ERR_INVALID_ARG;
    if (nlandmarks <= 128 || plane_stride < nlandmarks || row_stride < (split ? 2 : 5) * (int64_t)plane_stride || !(meas_var > 0.0f))
        return SLAM_OK;
    if (int rc = check_score_inputs(e, slot)) return rc;
    if (e->obs_nlandmarks < 0 || e->obs_nlandmarks != nlandmarks) return SLAM_ERR_NOT_READY;
    const int group = e->ekf_group_size(n, true, plane_stride, true, split != nullptr);
    if (!frame_front_fits(n, nlandmarks, group)) return SLAM_OK;
    HIP_TRY(e->ll_buf.ensure(sizeof(float) * (size_t)n));
    EkfArgs a;
    a.map_in = d_map_in;
    a.map_out = d_map_out;
    a.row_stride = row_stride;
    a.plane_stride = plane_stride;
    a.nlandmarks = nlandmarks;
    a.x = d_x;   // not read by the fused launch: the update works out its motion samples itself
    a.y = d_y;
    a.th = d_th;
    a.anc = split && split->map_anc ? split->map_anc : d_anc;
    a.n = n;
    a.obs_zx = e->d_obs_zx;
    a.obs_zy = e->d_obs_zy;
    a.meas_var = meas_var;
    a.loglik = e->ll_buf.as<float>();
    a.loglik_user = nullptr;
    a.xcd_chunk = 0;
    if (split) {
        a.group_filter = split->group_filter;
        a.cov = split->cov;
        a.cov_stride = split->cov_stride;
        a.covx = split->covx;
        a.covx_stride = split->covx_stride;
        a.cls_in = split->cls_in;
        a.cls_out = split->cls_out;
        a.cstamp = split->cstamp;
        a.stamp_now = split->stamp_now;
    }
    MotionIO io{ d_src_x, d_src_y, d_src_th, d_anc, d_x, d_y, d_th, FreeListRider() };
    int lanes = 0;
    // one bracket for the whole launch: it counts as the frame's landmark update (the dominant stage)
    ScoreGrid sg;
    if (int rc = many_pose_grid(e, slot, n, &sg)) return rc;
    HIP_TRY(launch_frame_front(e->stream, sg, e->d_bx, e->d_by, e->nbeams, io, first_id, dp, sigma, seed,
                               frame, d_score, d_count, a, group, e->prof_next(SLAM_PROF_EKF), launched, &lanes));
    if (*launched) {
        e->front_last[0] = group;
        e->front_last[1] = lanes;
        e->ll_n = n;
        e->ekf_form_launches[1]++;
        e->front_launches++;
    }
    return SLAM_OK;
}

int slam_ekf_split_dev(slam_engine* e, const float* d_mean_in, float* d_mean_out, int64_t row_stride, int plane_stride,
                       int nlandmarks, const float* d_x, const float* d_y, const float* d_th, const int32_t* d_anc, int n,
                       float meas_var, const slam::SplitIO* split)
{
    ENTER(e);
    if (n <= 0 || nlandmarks <= 0 || plane_stride < nlandmarks || row_stride < 2 * (int64_t)plane_stride || !(meas_var > 0.0f) ||
        !d_mean_in || !d_mean_out || d_mean_in == d_mean_out || !d_x || !d_y || !d_th || !split || !split->cov || !split->covx || !split->cls_in ||
        !split->cls_out || !split->cstamp)
        return SLAM_ERR_INVALID_ARG;
    if (e->obs_nlandmarks < 0 || e->obs_nlandmarks != nlandmarks) return SLAM_ERR_NOT_READY;
    HIP_TRY(e->ll_buf.ensure(sizeof(float) * (size_t)n));
    EkfArgs a;
    a.map_in = d_mean_in;
    a.map_out = d_mean_out;
    a.row_stride = row_stride;
    a.plane_stride = plane_stride;
    a.nlandmarks = nlandmarks;
    a.x = d_x;
    a.y = d_y;
    a.th = d_th;
    a.anc = d_anc;
    a.n = n;
    a.obs_zx = e->d_obs_zx;
    a.obs_zy = e->d_obs_zy;
    a.meas_var = meas_var;
    a.loglik = e->ll_buf.as<float>();
    a.loglik_user = nullptr;
    a.xcd_chunk = 0;
    a.group_filter = split->group_filter;
    a.cov = split->cov;
    a.cov_stride = split->cov_stride;
    a.covx = split->covx;
    a.covx_stride = split->covx_stride;
    a.cls_in = split->cls_in;
    a.cls_out = split->cls_out;
    a.cstamp = split->cstamp;
    a.stamp_now = split->stamp_now;
    const int group = e->ekf_group_size(n, d_anc != nullptr, plane_stride, false, true);
    HIP_TRY(launch_ekf_update(e->stream, a, e->prof_next(split->group_filter == 2 ? SLAM_PROF_EKF_TAIL : SLAM_PROF_EKF), group));
    if (split->group_filter != 2) e->ekf_form_launches[1]++;
    e->ll_n = n;
    return SLAM_OK;
}

int slam_frame_fusion_set(slam_engine* e, int on)
{
    ENTER(e);
    if (on < 0 || on > 1) return SLAM_ERR_INVALID_ARG;
    e->frame_fusion = on != 0;
    return SLAM_OK;
}

int slam_frame_fusion_count(slam_engine* e, int64_t* launches)
{
    ENTER(e);
    if (!launches) return SLAM_ERR_INVALID_ARG;
    *launches = e->front_launches;
    return SLAM_OK;
}

int slam_frame_front_last(slam_engine* e, int32_t info[2])
{
    ENTER(e);
    if (!info) return SLAM_ERR_INVALID_ARG;
    info[0] = e->front_last[0];
    info[1] = e->front_last[1];
    return SLAM_OK;
}

int slam_ekf_form_set(slam_engine* e, int form)
{
    ENTER(e);
    if (form < -1 || form > 2) return SLAM_ERR_INVALID_ARG;
    e->ekf_form = getenv("SLAM_EKF_GROUP") ? atoi(getenv("SLAM_EKF_GROUP")) : form;   // the environment wins (measurements)
    return SLAM_OK;
}

int slam_pf_paged_set(slam_engine* e, int on)
{
    ENTER(e);
    if (on < 0 || on > 1) return SLAM_ERR_INVALID_ARG;
    e->pf_paged = on != 0;
    return SLAM_OK;
}

int slam_ekf_inplace_form_set(slam_engine* e, int form)
{
    ENTER(e);
    if (form < -1 || form > 1) return SLAM_ERR_INVALID_ARG;
    e->ekf_inplace_form = getenv("SLAM_EKF_INPLACE") ? atoi(getenv("SLAM_EKF_INPLACE")) : form;
    return SLAM_OK;
}

int slam_ekf_inplace_form_counts(slam_engine* e, int64_t counts[2])
{
    ENTER(e);
    if (!counts) return SLAM_ERR_INVALID_ARG;
    counts[0] = e->ekf_inplace_launches[0];
    counts[1] = e->ekf_inplace_launches[1];
    return SLAM_OK;
}

int slam_ekf_form_counts(slam_engine* e, int64_t counts[2])
{
    ENTER(e);
    if (!counts) return SLAM_ERR_INVALID_ARG;
    counts[0] = e->ekf_form_launches[0];
    counts[1] = e->ekf_form_launches[1];
    return SLAM_OK;
}

static int logweight_common(slam_engine* e, const float* d_score, const float* d_loglik, float score_gain, int n,
                            float* d_logw, float* d_max, const CovArgs* cov = nullptr, int cov_bound = 0)
{
    if (n <= 0 || !d_logw) return SLAM_ERR_INVALID_ARG;
    if (e->bmax_buf.cap < sizeof(float) * (size_t)logweight_scratch_floats()) {   // block maxima + a ticket word kept at zero
        HIP_TRY(e->bmax_buf.ensure(sizeof(float) * (size_t)logweight_scratch_floats()));
        HIP_TRY(hipMemsetAsync(e->bmax_buf.p, 0, e->bmax_buf.cap, e->stream));
    }
    const ProfScope prof(e, SLAM_PROF_WEIGHTS);
    // with a resample gate: the weights of a frame that did not resample carry into this one (device-side decision)
    const bool carry = e->gate_frac_q16 != 0 && e->carry_n == n;
    HIP_TRY(launch_logweight(e->stream, d_score, d_loglik, score_gain, n, d_logw, e->bmax_buf.as<float>(), d_max,
                             carry ? e->carry_buf.as<float>() : nullptr, carry ? e->gate_buf.as<int32_t>() : nullptr, cov, cov_bound));
    e->bmax_count = logweight_scratch_elems(n);
    e->bmax_n = n;
    return SLAM_OK;
}

int slam_logweight_dev(slam_engine* e, const float* d_score, const float* d_loglik, float score_gain, int n,
                       float* d_logw, float* d_max)
{
    ENTER(e);
    return logweight_common(e, d_score, d_loglik, score_gain, n, d_logw, d_max);
}

int slam_logweight_ekf_dev(slam_engine* e, const float* d_score, float score_gain, int n, float* d_logw, float* d_max)
{
    ENTER(e);
    if (e->ll_n != n) return SLAM_ERR_NOT_READY;   // needs slam_ekf_update_dev(…, n, …) on this engine first
    return logweight_common(e, d_score, e->ll_buf.as<float>(), score_gain, n, d_logw, d_max);
}

// the session's form: d_loglik == nullptr -> the log-likelihoods the last landmark update left in the engine (use_ekf) or none;
// cov: a split session's covariance classes are brought up to date by workgroups of the same launch
int slam_logweight_cov_dev(slam_engine* e, const float* d_score, bool use_ekf, float score_gain, int n, float* d_logw, float* d_max,
                           const CovArgs* cov, int cov_bound)
{
    ENTER(e);
    if (use_ekf && e->ll_n != n) return SLAM_ERR_NOT_READY;
    return logweight_common(e, d_score, use_ekf ? e->ll_buf.as<float>() : nullptr, score_gain, n, d_logw, d_max, cov, cov_bound);
}

int slam_quantise_scan_dev(slam_engine* e, const float* d_logw, const float* d_max, int n, uint64_t* d_sum)
{
    ENTER(e);
    if (n <= 0 || !d_logw) return SLAM_ERR_INVALID_ARG;
    if (!d_max && e->bmax_n != n) return SLAM_ERR_NOT_READY;   // needs the maxima of slam_logweight_dev(n)
    const size_t ntiles = (size_t)scan_tile_count(n);
    HIP_TRY(e->scan_state.ensure(sizeof(uint64_t) * ((size_t)n + 3 * ntiles + 1)));
    uint64_t* cdf = e->scan_state.as<uint64_t>();
    uint64_t* tiles = cdf + n;   // tile_total | tile_s16 | tile_q16
    float* carry = nullptr;
    if (e->gate_frac_q16 != 0) {
        HIP_TRY(e->carry_buf.ensure(sizeof(float) * (size_t)n));
        carry = e->carry_buf.as<float>();
    }
    const ProfScope prof(e, SLAM_PROF_SCAN);
    HIP_TRY(launch_quantise_scan(e->stream, d_logw, d_max, e->bmax_buf.as<float>(), e->bmax_count, n, cdf, tiles, d_sum,
                                 carry, tiles + ntiles, tiles + 2 * ntiles, e->gate_buf.as<unsigned int>() + kGateTicketWord));
    e->scan_n = n;
    e->carry_n = carry ? n : -1;
    return SLAM_OK;
}

int slam_offspring_from_scan_dev(slam_engine* e, int n, const uint64_t* d_base, const uint64_t* d_total, uint64_t seed,
                                 uint32_t frame, int64_t n_total, int32_t* d_first)
{
    ENTER(e);
    if (n <= 0 || n_total < n || n_total > 0x7fffffff || !d_first) return SLAM_ERR_INVALID_ARG;
    if (e->scan_n != n) return SLAM_ERR_NOT_READY;
    const uint64_t* cdf = e->scan_state.as<uint64_t>();
    // (a shard of a larger population: base and total come from the caller, so does the gate — not applied here)
    HIP_TRY(launch_offspring_from_scan(e->stream, cdf, cdf + n, n, d_base, d_total, nullptr, 0, 1, seed, frame, n_total,
                                       d_first));
    return SLAM_OK;
}

int slam_ancestors_from_scan_dev(slam_engine* e, int n, uint64_t seed, uint32_t frame, int32_t* d_anc)
{
    ENTER(e);
    if (n <= 0 || !d_anc) return SLAM_ERR_INVALID_ARG;
    if (e->scan_n != n) return SLAM_ERR_NOT_READY;
    const uint64_t* cdf = e->scan_state.as<uint64_t>();
    const uint32_t frac = e->carry_n == n ? e->gate_frac_q16 : 0;   // the gate needs the sums of a gated quantise_scan
    const GateOut gate = frac ? e->gate_next() : GateOut();
    const ProfScope prof(e, SLAM_PROF_ANCESTORS);
    if (ancestors_from_scan_fits(n)) {
        // the distinct-ancestor count only steers the EKF's kernel choice: made only for populations that have maps
        HIP_TRY(launch_ancestors_from_scan(e->stream, cdf, cdf + n, n, seed, frame, d_anc, frac, gate,
                                           e->ll_n == n ? e->heads_out() : HeadsOut()));
        return SLAM_OK;
    }
    // more tiles than the one-launch form keeps in LDS: the two-launch form through a scratch `first` array
    HIP_TRY(e->first_buf.ensure(sizeof(int32_t) * (size_t)n));
    int32_t* first = e->first_buf.as<int32_t>();
    HIP_TRY(launch_offspring_from_scan(e->stream, cdf, cdf + n, n, nullptr, nullptr, nullptr, 0, 1, seed, frame, n, first,
                                       frac, gate));
    HIP_TRY(launch_ancestors(e->stream, first, n, 0, n, d_anc));
    return SLAM_OK;
}

int slam_offspring_from_scan_sharded_dev(slam_engine* e, int n, const uint64_t* d_shard_totals, int rank, int world,
                                         uint64_t seed, uint32_t frame, int64_t n_total, int32_t* d_first)
{
    ENTER(e);
    if (n <= 0 || world < 1 || world > kMaxRanks || rank < 0 || rank >= world || n_total != (int64_t)n * world ||
        n_total > 0x7fffffff || !d_shard_totals || !d_first)
        return SLAM_ERR_INVALID_ARG;
    if (e->scan_n != n) return SLAM_ERR_NOT_READY;
    const uint64_t* cdf = e->scan_state.as<uint64_t>();
    const uint32_t frac = e->carry_n == n ? e->gate_frac_q16 : 0;
    const ProfScope prof(e, SLAM_PROF_ANCESTORS);
    HIP_TRY(launch_offspring_from_scan(e->stream, cdf, cdf + n, n, nullptr, nullptr, d_shard_totals, rank, world, seed,
                                       frame, n_total, d_first, frac, frac ? e->gate_next() : GateOut()));
    return SLAM_OK;
}

int slam_resample_gate_set(slam_engine* e, float ess_frac)
{
    ENTER(e);
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->gate_frac_q16 = ess_frac > 0.0f && ess_frac < 1.0f ? (uint32_t)lrintf(ess_frac * 65536.0f) : 0u;
    e->carry_n = -1;
    e->h_gate[0] = 1;
    const int32_t one = 1;   // nothing is carried into the next frame
    HIP_TRY(hipMemcpy(e->gate_buf.p, &one, sizeof one, hipMemcpyHostToDevice));
    return SLAM_OK;
}

int slam_resample_happened_host(slam_engine* e, int* resampled)
{
    ENTER(e);
    if (!resampled) return SLAM_ERR_INVALID_ARG;
    *resampled = 1;
    if (e->gate_frac_q16 == 0 || e->gate_seq == 0) return SLAM_OK;   // no gate (or no gated stage yet): every frame resamples
    volatile uint32_t* h_seq = reinterpret_cast<volatile uint32_t*>(e->h_gate + 1);
    const uint32_t seq = e->gate_seq;
    if (e->comm) {   // sharded: the verdict sits behind collectives
        if (int rc = comm_wait_flag(e->comm, h_seq, seq)) return rc;
        *resampled = e->h_gate[0] != 0;
        return SLAM_OK;
    }
    bool arrived = false;
    for (long spin = 0; spin < 400000000L; ++spin) {   // bounded: a few seconds at most
        if (__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) == seq) { arrived = true; break; }
    }
    if (!arrived) {
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) != seq) return fail_hip(e, hipErrorUnknown, "resample gate flag");
    }
    *resampled = e->h_gate[0] != 0;
    return SLAM_OK;
}

int slam_quantise_weights_dev(slam_engine* e, const float* d_logw, const float* d_max, int n, uint64_t* d_wq,
                              uint64_t* d_sum)
{
    ENTER(e);
    if (n < 0 || !d_max || !d_sum || (n > 0 && (!d_logw || !d_wq))) return SLAM_ERR_INVALID_ARG;
    HIP_TRY(launch_quantise_weights(e->stream, d_logw, d_max, n, d_wq, d_sum));
    return SLAM_OK;
}

int slam_prefix_sum_dev(slam_engine* e, const uint64_t* d_wq, int n, uint64_t* d_cdf)
{
    ENTER(e);
    if (n < 0 || (n > 0 && (!d_wq || !d_cdf))) return SLAM_ERR_INVALID_ARG;
    if (n == 0) return SLAM_OK;
    HIP_TRY(e->scratch.ensure(sizeof(uint64_t) * (size_t)prefix_sum_scratch_elems(n)));
    HIP_TRY(launch_prefix_sum(e->stream, d_wq, n, d_cdf, e->scratch.as<uint64_t>()));
    return SLAM_OK;
}

int slam_offspring_offsets_dev(slam_engine* e, const uint64_t* d_cdf, int n, const uint64_t* d_base,
                               const uint64_t* d_total, uint64_t seed, uint32_t frame, int64_t n_total,
                               int32_t* d_first)
{
    ENTER(e);
    if (n < 0 || n_total < n || n_total > 0x7fffffff || !d_total || (n > 0 && (!d_cdf || !d_first)))
        return SLAM_ERR_INVALID_ARG;
    HIP_TRY(launch_offspring_offsets(e->stream, d_cdf, n, d_base, d_total, seed, frame, n_total, d_first));
    return SLAM_OK;
}

int slam_ancestors_dev(slam_engine* e, const int32_t* d_first_all, int64_t n_total, int64_t slot0, int nslots,
                       int32_t* d_anc)
{
    ENTER(e);
    if (nslots < 0 || n_total <= 0 || slot0 < 0 || slot0 + nslots > n_total || !d_first_all || (nslots > 0 && !d_anc))
        return SLAM_ERR_INVALID_ARG;
    HIP_TRY(launch_ancestors(e->stream, d_first_all, n_total, slot0, nslots, d_anc));
    return SLAM_OK;
}

uint64_t slam_comb_offset(uint64_t seed, uint32_t frame, uint64_t total)
{
    uint32_t c[4] = { 0u, 0u, frame, 1u /* resample stream */ };
    philox_host(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint64_t r64 = (uint64_t)c[0] | ((uint64_t)c[1] << 32);
    return (uint64_t)(((unsigned __int128)r64 * total) >> 64);
}

static bool make_plan(MigratePlan& plan, const int64_t* lo, const int32_t* cnt, int world)
{
    if (world < 1 || world > kMaxRanks || !lo || !cnt) return false;
    plan.world = world;
    int64_t off = 0;
    for (int q = 0; q < world; ++q) {
        if (cnt[q] < 0 || lo[q] < 0) return false;
        plan.lo[q] = lo[q];
        plan.off[q] = (int32_t)off;
        off += cnt[q];
        if (off > 0x7fffffff) return false;
    }
    plan.off[world] = (int32_t)off;
    return true;
}

int slam_ancestors_sharded_dev(slam_engine* e, const int32_t* d_first_all, int64_t n_total, int n_local, int rank,
                               int world, int32_t* d_src, int32_t* d_plan, int32_t* d_pose_idx)
{
    ENTER(e);
    if (n_local <= 0 || world < 1 || world > kMaxRanks || rank < 0 || rank >= world ||
        n_total != (int64_t)n_local * world || !d_first_all || !d_src || !d_plan ||
        (d_pose_idx && 3 * n_total > 0x7fffffff))
        return SLAM_ERR_INVALID_ARG;
    HIP_TRY(e->shard_buf.ensure(sizeof(int32_t) * (size_t)shard_scan_words(n_local)));
    const uint32_t seq = ++e->plan_seq;
    const ProfScope prof(e, SLAM_PROF_PLAN);
    HIP_TRY(launch_ancestors_sharded(e->stream, d_first_all, n_total, n_local, rank, world, e->shard_buf.as<int32_t>(),
                                     d_plan, d_src, d_pose_idx, e->d_hplan,
                                     reinterpret_cast<uint32_t*>(e->d_hplan + SLAM_PLAN_WORDS(kMaxRanks)), seq, e->exch_cap,
                                     e->d_hheads));
    e->shard_n = n_local;   // what slam_migrate_pack_dev will read
    e->plan_world = world;
    return SLAM_OK;
}

int slam_exchange_set_capacity(slam_engine* e, int recv_capacity)
{
    ENTER(e);
    e->exch_cap = recv_capacity > 0 ? recv_capacity : 0x7fffffff;
    return SLAM_OK;
}

int slam_exchange_plan_host(slam_engine* e, int world, int32_t* plan)
{
    ENTER(e);
    if (!plan || world < 1 || world > kMaxRanks) return SLAM_ERR_INVALID_ARG;
    if (e->plan_seq == 0 || e->plan_world != world) return SLAM_ERR_NOT_READY;
    volatile uint32_t* h_flag = reinterpret_cast<volatile uint32_t*>(e->h_plan + SLAM_PLAN_WORDS(kMaxRanks));
    const uint32_t seq = e->plan_seq;
    if (e->comm) {   // the plan kernel sits behind this frame's collectives: poll the communicator while waiting, bounded in time
        if (int rc = comm_wait_flag(e->comm, h_flag, seq)) return rc;
    } else {
        bool arrived = false;
        for (long spin = 0; spin < 400000000L; ++spin) {   // bounded: a few seconds at most
            if (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) == seq) { arrived = true; break; }
        }
        if (!arrived) {   // the launch failed or the device is wedged: let the runtime tell us
            HIP_TRY(hipStreamSynchronize(e->stream));
            if (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != seq) return fail_hip(e, hipErrorUnknown, "exchange plan flag");
        }
    }
    memcpy(plan, e->h_plan, sizeof(int32_t) * (size_t)SLAM_PLAN_WORDS(world));
    return SLAM_OK;
}

int slam_migrate_pack_dev(slam_engine* e, int n_local, int rank, int world, const int32_t* plan, const float* d_pose,
                          int64_t pose_ld, const float* d_map, int64_t row_stride, int plane_stride, int nlandmarks,
                          float* d_out)
{
    return slam_migrate_pack_paged(e, n_local, rank, world, plan, d_pose, pose_ld, d_map, row_stride, plane_stride, nlandmarks,
                                   d_out, nullptr, 0, nullptr, nullptr, nullptr);
}

// d_pt != nullptr: d_map is a page pool and the particles' landmarks sit behind page tables of nb entries (pf_session.hip)
int slam_migrate_pack_paged(slam_engine* e, int n_local, int rank, int world, const int32_t* plan, const float* d_pose,
                            int64_t pose_ld, const float* d_map, int64_t row_stride, int plane_stride, int nlandmarks,
                            float* d_out, const int32_t* d_pt, int nb, const float* d_split_cov, const int32_t* d_split_cls,
                            const slam::PageGeom* geom)
{
    ENTER(e);
    if (n_local <= 0 || world < 1 || world > kMaxRanks || rank < 0 || rank >= world || !plan || nlandmarks < 0 ||
        !d_pose || (nlandmarks > 0 && (!d_map || plane_stride < nlandmarks || row_stride < (d_split_cls ? 2 : 5) * (int64_t)plane_stride)) ||
        (d_split_cls && !d_split_cov) || (d_split_cls && d_pt && !geom))
        return SLAM_ERR_INVALID_ARG;
    if (e->shard_n != n_local) return SLAM_ERR_NOT_READY;   // needs slam_ancestors_sharded_dev(n_local) of this frame
    MigratePlan mp;
    int64_t base[kMaxRanks];
    for (int q = 0; q < world; ++q) base[q] = plan[1 + 2 * world + q];
    if (!make_plan(mp, base, plan + 1, world) || plan[1 + rank] != 0) return SLAM_ERR_INVALID_ARG;
    if (mp.off[world] > 0 && !d_out) return SLAM_ERR_INVALID_ARG;
    const ProfScope prof(e, SLAM_PROF_PACK);
    HIP_TRY(launch_migrate_pack(e->stream, e->shard_buf.as<int32_t>(), n_local, mp, d_pose, pose_ld, d_map, row_stride,
                                plane_stride, nlandmarks, d_out, d_pt, nb, d_split_cov, d_split_cls, geom ? *geom : PageGeom()));
    return SLAM_OK;
}

int slam_migrate_unpack_dev(slam_engine* e, const float* d_in, int world, const int32_t* recv_cnt, int n_local,
                            float* d_pose, int64_t pose_ld, float* d_map, int64_t row_stride, int plane_stride,
                            int nlandmarks)
{
    ENTER(e);
    MigratePlan plan;
    int64_t zeros[kMaxRanks] = { 0 };
    if (!make_plan(plan, zeros, recv_cnt, world) || n_local <= 0 || nlandmarks < 0 || !d_pose ||
        (nlandmarks > 0 && (!d_map || plane_stride < nlandmarks || row_stride < 5 * (int64_t)plane_stride)))
        return SLAM_ERR_INVALID_ARG;
    if (plan.off[world] > 0 && !d_in) return SLAM_ERR_INVALID_ARG;
    if ((int64_t)n_local + plan.off[world] > pose_ld) return SLAM_ERR_CAPACITY;   // pose_ld = particle capacity
    const ProfScope prof(e, SLAM_PROF_UNPACK);
    HIP_TRY(launch_migrate_unpack(e->stream, d_in, plan, n_local, d_pose, pose_ld, d_map, row_stride, plane_stride,
                                  nlandmarks));
    return SLAM_OK;
}

int slam_argmax_dev(slam_engine* e, const float* d_values, int n, int32_t* d_index, float* d_value)
{
    ENTER(e);
    if (n <= 0 || !d_values || !d_index || !d_value) return SLAM_ERR_INVALID_ARG;
    HIP_TRY(launch_argmax(e->stream, d_values, n, d_index, d_value));
    return SLAM_OK;
}

int slam_gather_f32_dev(slam_engine* e, const float* d_src, const int32_t* d_idx, int n, float* d_dst)
{
    ENTER(e);
    if (n < 0 || (n > 0 && (!d_src || !d_idx || !d_dst)) || d_src == d_dst) return SLAM_ERR_INVALID_ARG;
    HIP_TRY(launch_gather_f32(e->stream, d_src, d_idx, n, d_dst));
    return SLAM_OK;
}

int slam_gather_map_dev(slam_engine* e, const float* d_map_in, float* d_map_out, int64_t in_row_stride,
                        int64_t out_row_stride, int in_plane_stride, int out_plane_stride, int nlandmarks,
                        const int32_t* d_idx, int n)
{
    ENTER(e);
    if (n < 0 || nlandmarks < 0 || in_plane_stride < nlandmarks || out_plane_stride < nlandmarks ||
        in_row_stride < 5 * (int64_t)in_plane_stride || out_row_stride < 5 * (int64_t)out_plane_stride ||
        (n > 0 && nlandmarks > 0 && (!d_map_in || !d_map_out || !d_idx)) || d_map_in == d_map_out)
        return SLAM_ERR_INVALID_ARG;
    HIP_TRY(launch_gather_map(e->stream, d_map_in, d_map_out, in_row_stride, out_row_stride, in_plane_stride,
                              out_plane_stride, nlandmarks, d_idx, n));
    return SLAM_OK;
}

}  // extern "C"
