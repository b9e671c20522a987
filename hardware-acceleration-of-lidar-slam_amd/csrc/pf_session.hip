// pf_session.hip — the slam_pf_* convenience object of include/slam_hip.h: device buffers + one frame per
// call, built entirely on the public stage entry points (so it runs exactly what pf.py runs on one GPU).
// No counterpart in the reference; specification: oracle/slam_oracle_pf.c.

#include <hip/hip_runtime.h>
#include <string.h>

#include <new>
#include <vector>

#include "kernels.h"

using namespace slam;

struct slam_pf {
    slam_engine* e = nullptr;
    slam_pf_config cfg{};
    int n = 0, L = 0;
    float* pose[2] = { nullptr, nullptr };     // [3][n] each
    float* map[2] = { nullptr, nullptr };      // [n][5][Lp] each: one row per particle, planes padded to Lp floats
    int Lp = 0;                                // plane stride: L rounded up to 32 floats (128-byte rows)
    int32_t* anc[2] = { nullptr, nullptr };
    float *score = nullptr, *logw = nullptr;
    int32_t *count = nullptr, *first = nullptr, *best_idx = nullptr;
    float* best_val = nullptr;
    int cur = 0;       // pose / ancestor buffer holding the current particles
    int map_cur = 0;   // map buffer holding the current maps (flips only when the maps are rewritten)
    bool has_anc = false;
    uint32_t frame = 0;
};

namespace {

// the engine's stream, obtained through the public API only
hipError_t dev_alloc(void** p, size_t bytes) { return hipMalloc(p, bytes ? bytes : 4); }

int gathered_copy_out(slam_pf* pf, const float* d_src, float* h_dst, float* d_tmp)
{
    // d_src: n floats; apply the pending gather on the device, then copy back
    if (pf->has_anc) {
        int rc = slam_gather_f32_dev(pf->e, d_src, pf->anc[pf->cur], pf->n, d_tmp);
        if (rc != SLAM_OK) return rc;
        d_src = d_tmp;
    }
    int rc = slam_engine_sync(pf->e);
    if (rc != SLAM_OK) return rc;
    return hipMemcpy(h_dst, d_src, sizeof(float) * (size_t)pf->n, hipMemcpyDeviceToHost) == hipSuccess ? SLAM_OK
                                                                                                      : SLAM_ERR_HIP;
}

}  // namespace

extern "C" {

int slam_pf_create(slam_engine* e, const slam_pf_config* cfg, slam_pf** out)
{
    if (!e || !cfg || !out || cfg->n_particles <= 0 || cfg->n_landmarks < 0 || !(cfg->meas_var > 0.0f))
        return SLAM_ERR_INVALID_ARG;
    *out = nullptr;
    if (int rc = slam_engine_sync(e)) return rc;   // also selects the engine's device
    slam_pf* pf = new (std::nothrow) slam_pf();
    if (!pf) return SLAM_ERR_HIP;
    pf->e = e;
    pf->cfg = *cfg;
    pf->n = cfg->n_particles;
    pf->L = cfg->n_landmarks;
    pf->Lp = (pf->L + 31) / 32 * 32;
    const size_t n = (size_t)pf->n, L = (size_t)pf->L, Lp = (size_t)pf->Lp;
    bool ok = true;
    for (int b = 0; b < 2; ++b) {
        ok = ok && dev_alloc((void**)&pf->pose[b], 3 * n * 4) == hipSuccess;
        ok = ok && dev_alloc((void**)&pf->anc[b], n * 4) == hipSuccess;
        if (L) ok = ok && dev_alloc((void**)&pf->map[b], 5 * Lp * n * 4) == hipSuccess;
    }
    ok = ok && dev_alloc((void**)&pf->score, n * 4) == hipSuccess && dev_alloc((void**)&pf->logw, n * 4) == hipSuccess &&
         dev_alloc((void**)&pf->count, n * 4) == hipSuccess && dev_alloc((void**)&pf->first, n * 4) == hipSuccess &&
         dev_alloc((void**)&pf->best_idx, 4) == hipSuccess && dev_alloc((void**)&pf->best_val, 4) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        slam_pf_destroy(pf);
        return SLAM_ERR_HIP;
    }
    *out = pf;
    const float origin[3] = { 0, 0, 0 };
    return slam_pf_reset(pf, origin);
}

int slam_pf_destroy(slam_pf* pf)
{
    if (!pf) return SLAM_OK;
    (void)slam_engine_sync(pf->e);
    for (int b = 0; b < 2; ++b) {
        (void)hipFree(pf->pose[b]);
        (void)hipFree(pf->anc[b]);
        (void)hipFree(pf->map[b]);
    }
    (void)hipFree(pf->score);
    (void)hipFree(pf->logw);
    (void)hipFree(pf->count);
    (void)hipFree(pf->first);
    (void)hipFree(pf->best_idx);
    (void)hipFree(pf->best_val);
    delete pf;
    return SLAM_OK;
}

int slam_pf_reset(slam_pf* pf, const float pose[3])
{
    if (!pf || !pose) return SLAM_ERR_INVALID_ARG;
    const size_t n = (size_t)pf->n;
    std::vector<float> h(3 * n);
    for (int k = 0; k < 3; ++k)
        for (size_t i = 0; i < n; ++i) h[k * n + i] = pose[k];
    if (int rc = slam_engine_sync(pf->e)) return rc;
    if (hipMemcpy(pf->pose[pf->cur], h.data(), 3 * n * 4, hipMemcpyHostToDevice) != hipSuccess) return SLAM_ERR_HIP;
    if (pf->L) {   // P_xx = -1: "not seen yet"
        const size_t Lp = (size_t)pf->Lp;
        std::vector<float> m(5 * Lp * n, 0.0f);
        for (size_t i = 0; i < n; ++i)
            for (size_t l = 0; l < Lp; ++l) m[(5 * i + 2) * Lp + l] = -1.0f;
        if (hipMemcpy(pf->map[pf->map_cur], m.data(), m.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
            return SLAM_ERR_HIP;
    }
    pf->has_anc = false;
    pf->frame = 0;
    return SLAM_OK;
}

int slam_pf_set_poses_host(slam_pf* pf, const float* x, const float* y, const float* theta)
{
    if (!pf || !x || !y || !theta) return SLAM_ERR_INVALID_ARG;
    const size_t n = (size_t)pf->n;
    if (int rc = slam_engine_sync(pf->e)) return rc;
    float* d = pf->pose[pf->cur];
    if (hipMemcpy(d, x, n * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + n, y, n * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + 2 * n, theta, n * 4, hipMemcpyHostToDevice) != hipSuccess)
        return SLAM_ERR_HIP;
    pf->has_anc = false;
    return SLAM_OK;
}

int slam_pf_set_map_host(slam_pf* pf, const float* rows)
{
    if (!pf || !rows || !pf->L) return SLAM_ERR_INVALID_ARG;
    if (int rc = slam_engine_sync(pf->e)) return rc;
    if (pf->has_anc) return SLAM_ERR_NOT_READY;   // set poses / reset first: a gather is pending
    // host [n][5][L] -> device [n][5][Lp]: 5n planes of L floats each
    if (hipMemcpy2D(pf->map[pf->map_cur], (size_t)pf->Lp * 4, rows, (size_t)pf->L * 4, (size_t)pf->L * 4,
                    5 * (size_t)pf->n, hipMemcpyHostToDevice) != hipSuccess)
        return SLAM_ERR_HIP;
    return SLAM_OK;
}

int slam_pf_step(slam_pf* pf, int slot, const float dp[3], int use_observations)
{
    if (!pf || !dp) return SLAM_ERR_INVALID_ARG;
    slam_engine* e = pf->e;
    const int n = pf->n, L = pf->L, cur = pf->cur, nxt = 1 - cur;
    const size_t sn = (size_t)n;
    const float* src = pf->pose[cur];
    float* dst = pf->pose[nxt];
    const int32_t* anc = pf->has_anc ? pf->anc[cur] : nullptr;
    int rc = slam_motion_score_dev(e, slot, src, src + sn, src + 2 * sn, anc, dst, dst + sn, dst + 2 * sn, n, 0, dp,
                                   pf->cfg.sigma, pf->cfg.seed, pf->frame, pf->score, pf->count);
    if (rc != SLAM_OK) return rc;
    const bool ekf = L > 0 && use_observations;
    const int mc = pf->map_cur, mn = 1 - mc;
    if (ekf) {
        rc = slam_ekf_update_dev(e, pf->map[mc], pf->map[mn], 5 * (int64_t)pf->Lp, pf->Lp, L, dst, dst + sn, dst + 2 * sn, anc, n,
                                 pf->cfg.meas_var, nullptr);
        if (rc != SLAM_OK) return rc;
        pf->map_cur = mn;
        rc = slam_logweight_ekf_dev(e, pf->score, pf->cfg.score_gain, n, pf->logw, nullptr);
    } else {
        if (L > 0 && anc) {   // the maps follow their particles even without an observation
            rc = slam_gather_map_dev(e, pf->map[mc], pf->map[mn], 5 * (int64_t)pf->Lp, 5 * (int64_t)pf->Lp, pf->Lp, pf->Lp, L,
                                     anc, n);
            if (rc != SLAM_OK) return rc;
            pf->map_cur = mn;
        }
        rc = slam_logweight_dev(e, pf->score, nullptr, pf->cfg.score_gain, n, pf->logw, nullptr);
    }
    if (rc != SLAM_OK) return rc;
    if ((rc = slam_quantise_scan_dev(e, pf->logw, nullptr, n, nullptr)) != SLAM_OK) return rc;
    if ((rc = slam_ancestors_from_scan_dev(e, n, pf->cfg.seed, pf->frame, pf->anc[nxt])) != SLAM_OK) return rc;
    pf->cur = nxt;
    pf->has_anc = true;
    pf->frame++;
    return SLAM_OK;
}

int slam_pf_best(slam_pf* pf, float pose[3], float* logw, int32_t* index)
{
    if (!pf || !pose) return SLAM_ERR_INVALID_ARG;
    // the log-weights of the last frame belong to pose[cur] BEFORE the pending gather
    int rc = slam_argmax_dev(pf->e, pf->logw, pf->n, pf->best_idx, pf->best_val);
    if (rc != SLAM_OK) return rc;
    if ((rc = slam_engine_sync(pf->e)) != SLAM_OK) return rc;
    int32_t idx = 0;
    float val = 0;
    if (hipMemcpy(&idx, pf->best_idx, 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&val, pf->best_val, 4, hipMemcpyDeviceToHost) != hipSuccess)
        return SLAM_ERR_HIP;
    const float* p = pf->pose[pf->cur];
    for (int k = 0; k < 3; ++k)
        if (hipMemcpy(&pose[k], p + (size_t)k * pf->n + idx, 4, hipMemcpyDeviceToHost) != hipSuccess) return SLAM_ERR_HIP;
    if (logw) *logw = val;
    if (index) *index = idx;
    return SLAM_OK;
}

int slam_pf_get_poses_host(slam_pf* pf, float* x, float* y, float* theta)
{
    if (!pf || !x || !y || !theta) return SLAM_ERR_INVALID_ARG;
    const size_t n = (size_t)pf->n;
    const float* p = pf->pose[pf->cur];
    float* tmp = pf->pose[1 - pf->cur];   // the other buffer is free between frames
    float* out[3] = { x, y, theta };
    for (int k = 0; k < 3; ++k)
        if (int rc = gathered_copy_out(pf, p + k * n, out[k], tmp)) return rc;
    return SLAM_OK;
}

int slam_pf_get_map_host(slam_pf* pf, float* rows)
{
    if (!pf || !rows || !pf->L) return SLAM_ERR_INVALID_ARG;
    const size_t n = (size_t)pf->n, L = (size_t)pf->L, Lp = (size_t)pf->Lp;
    const float* src = pf->map[pf->map_cur];
    if (pf->has_anc) {
        int rc = slam_gather_map_dev(pf->e, pf->map[pf->map_cur], pf->map[1 - pf->map_cur], 5 * (int64_t)Lp,
                                     5 * (int64_t)Lp, pf->Lp, pf->Lp, pf->L, pf->anc[pf->cur], pf->n);
        if (rc != SLAM_OK) return rc;
        src = pf->map[1 - pf->map_cur];
    }
    if (int rc = slam_engine_sync(pf->e)) return rc;
    return hipMemcpy2D(rows, L * 4, src, Lp * 4, L * 4, 5 * n, hipMemcpyDeviceToHost) == hipSuccess ? SLAM_OK : SLAM_ERR_HIP;
}

}  // extern "C"
