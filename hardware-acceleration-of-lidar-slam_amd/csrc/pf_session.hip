// pf_session.hip — the slam_pf_* session of include/slam_hip.h: device buffers + one whole frame per call, for a
// plain C host.  One GPU (slam_pf_create) or one rank of a population sharded over several GPUs
// (slam_pf_create_sharded): the frame is built on the public stage entry points either way, and in the sharded
// form every exchange step between the ranks is issued from here through comm.h (RCCL over xGMI, or the
// in-process transport) — nothing but this file sits between the launches.
// No counterpart in the reference; specification: oracle/slam_oracle_pf.c; the shape
// "handle created once in main and threaded through" is the reference's (Hadrware_acclereated.cpp:842-845, 284).

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "comm.h"
#include "engine_internal.h"
#include "kernels.h"

using namespace slam;

struct slam_pf {
    slam_engine* e = nullptr;
    slam_pf_config cfg{};
    int n = 0, L = 0;
    float* pose[2] = { nullptr, nullptr };     // [3][n] each
    float* map[2] = { nullptr, nullptr };      // [cap][5][Lp] each: one row per particle, planes padded to Lp floats
    int Lp = 0;                                // plane stride: L rounded up to 32 floats (128-byte rows)
    int32_t* anc[2] = { nullptr, nullptr };
    float *score = nullptr, *logw = nullptr;
    int32_t *count = nullptr, *first = nullptr;
    int cur = 0;       // pose / ancestor buffer holding the current particles
    int map_cur = 0;   // map buffer holding the current maps (flips only when the maps are rewritten)
    bool has_anc = false;
    uint32_t frame = 0;

    // ---- sharded form (comm != nullptr): this rank holds particles [rank*n, (rank+1)*n) of world*n
    slam_comm* comm = nullptr;
    int rank = 0, world = 1, recv_cap = 0, cap = 0;   // cap = n + recv_cap rows per map buffer (staging tail)
    int64_t n_total = 0;
    float* pose_all = nullptr;      // [world][x|y|theta][n]: every rank's poses, all-gathered each frame beside the EKF
    float* pose_stage = nullptr;    // [3][cap]: where the poses inside migrated records land (nothing reads them)
    int32_t* pose_idx[2] = { nullptr, nullptr };   // position of every slot's ancestor in pose_all
    int32_t* first_all = nullptr;   // [n_total]
    float* d_max = nullptr;         // weight normaliser (all-reduced)
    uint64_t *d_sum = nullptr, *totals = nullptr;   // shard total; all-gathered shard totals [world]
    int32_t* d_plan = nullptr;      // exchange plan of the frame, device copy
    float *sbuf = nullptr, *rbuf = nullptr;   // grow-only exchange buffers
    size_t sbuf_floats = 0, rbuf_floats = 0;
    bool exchange_pending = false;  // resample done, map rows not exchanged yet
    int rows_received = 0;
    // results a host asks for every frame (heaviest particle, posterior mean): written by ONE kernel to mapped host
    // memory behind a sequence number — no device-to-host copies, no stream synchronisation
    float* h_res = nullptr;         // pinned + mapped: 8 x 8 bytes of payload, the sequence number (word 16), and word 20:
                                    // "a free list came out shorter than its reservation" (paged maps; see free_list_kernel)
    float* d_hres = nullptr;        // the same memory as the device sees it
    uint32_t res_seq = 0;
    float* res_dev = nullptr;       // device copy of the payload (what the ranks all-gather)
    unsigned long long* sums_acc = nullptr;   // 4 accumulators + ticket of pose_sums_kernel (kept zeroed by the kernel)
    void* res_all = nullptr;        // [world] payloads
    // ---- paged maps (slam_pf_paged_set before the session is made; one GPU): copy-on-write pages behind a page table
    // per particle instead of one row per particle (paged_kernels.hip); map[] stays unallocated
    bool paged = false;
    int nb = 0, npages = 0;         // pages per particle; pages in the pool (2 * cap * nb: never fewer than half are free)
    float* store = nullptr;         // the one allocation behind map[0], map[1] and pool
    int32_t* votes = nullptr;       // [2] SLAM_MAP_AUTO's sample counters on the device
    float* pool = nullptr;          // [npages][5][32]
    int32_t* pt[2] = { nullptr, nullptr };   // [n][nb] page tables, current and next
    int pt_cur = 0;
    int32_t* freelist = nullptr;    // [npages] ascending free pages as of the last update
    uint32_t* stamp = nullptr;      // [npages] frame stamp of the last table that named the page
    uint32_t stamp_now = 0;
    int32_t* page_scratch = nullptr;   // the free list's bookkeeping (pool_state_words()) | count | tpage[nb] | tindex[nb] | tmask[nb] | tbase[nb + 1] |
                                       // the frame's observation list id[Lp] | zx[Lp] | zy[Lp] | round[Lp] | {count, highest round}
    // ---- split layout (SLAM_MAP_SPLIT, and what SLAM_MAP_AUTO keeps a single-GPU session on while its frames observe most
    // landmarks): means per particle, covariances per covariance class (split_kernels.hip); carved out of the same store
    bool split = false;
    bool dense_split = false;       // AUTO's layout for dense frames is split (else rows)
    float* mean[2] = { nullptr, nullptr };   // [cap][2][Lp]: mean[0] and cov share one half of the store, mean[1] starts the other
    float* cov = nullptr;           // [cap][3][Lp], updated in place once per class and frame
    float* covx = nullptr;          // [cap][2][Lp]: the determinant terms of the same covariances (behind mean[1])
    int sp_base = 0;                // the half of the store that holds mean[0] and cov
    int sp_cur = 0;                 // mean / class buffer of the current particles
    int32_t* cls[2] = { nullptr, nullptr };    // [cap]
    int32_t* live[2] = { nullptr, nullptr };   // [cap] the classes in use, current list and next
    int32_t* cov_cnt = nullptr;     // [3] list lengths, rotating (see cov_update_kernel)
    uint32_t* cstamp = nullptr;     // [cap]
    uint32_t cstamp_now = 0, cls_epoch = 0;
    int live_cur = 0, cov_phase = 0;
    // sharded: a row that arrives from another rank gets a class of its own; the numbers come from a free list on the device
    // (split_kernels.hip: class_free_list_kernel), handed out by the host: a fresh list holds at least recv_cap numbers
    int32_t* cls_free = nullptr;    // [cap]
    int32_t* cls_fs = nullptr;      // two words of the list kernel's bookkeeping
    int64_t cls_cursor = 0;         // entries of the current list handed out so far (beyond its guaranteed length: make a new one)
    uint32_t cls_appended = 0;      // classes appended to the list so far in this epoch (what cov_update_kernel's `mark` carries)
    void* split_scratch = nullptr;  // flags, prefix sums of a rows -> split move
    bool gated = false;             // cfg.resample_ess_frac in (0, 1): a frame resamples only when its ESS is low
    int64_t frames_resampled = 0;   // (as far as the host has looked: one frame behind)
    // SLAM_MAP_AUTO: the session watches how many landmarks the frames observe ({observed, L, seq} in words 24..26 of h_res,
    // left there by page_list_kernel / obs_count_kernel) and moves between rows and pages while it runs
    int layout_cfg = SLAM_MAP_AUTO;
    uint32_t obs_seq_issued = 0, obs_seq_seen = 0;
    int votes_pages = 0, votes_rows = 0;
    bool auto_stuck = false;        // a conversion ran out of memory: stay where we are
    int64_t conversions = 0;
    bool counted = false;           // this session holds the engine's one session slot
    bool last_ekf = false;          // the last frame ran the landmark update (its log-likelihoods are in the engine)
    int32_t* sel = nullptr;         // grow-only scratch of slam_pf_get_map_rows_host: chosen particles | their source rows
    int sel_cap = 0;
    float* conv_tmp = nullptr;      // grow-only scratch of a pages -> rows move (convert_to_rows)
    size_t conv_floats = 0;
};

namespace {

hipError_t dev_alloc(void** p, size_t bytes) { return hipMalloc(p, bytes ? bytes : 4); }

int convert_rows_to_split(slam_pf* pf);

int grow(slam_pf* pf, float** buf, size_t* have, size_t want)
{
    if (want <= *have) return SLAM_OK;
    slam_engine* e = pf->e;
    if (pf->comm) {   // an exchange still in flight may read the old buffer
        if (int rc = comm_wait_stream(pf->comm)) return rc;
    } else {
        SLAM_HIP_TRY(e, hipStreamSynchronize(e->stream));
    }
    const size_t cap = want > 2 * *have ? want + want / 2 : 2 * *have;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    SLAM_HIP_TRY(e, hipMalloc((void**)buf, cap * sizeof(float)));
    *have = cap;
    return SLAM_OK;
}

float* split_pool(const slam_pf* pf);
PageGeom split_geom(const slam_pf* pf);

// Map rows (and poses) of ancestors that live on another rank -> the staging tail of the current buffers, where
// the next EKF's fused gather picks them up.  pack (one launch) -> one grouped send/recv -> unpack (one launch).
// A remote ancestor travels once per destination rank, however many slots there descend from it.
int migrate(slam_pf* pf)
{
    slam_engine* e = pf->e;
    const int G = pf->world, n = pf->n, L = pf->L;
    int32_t plan[SLAM_PLAN_WORDS(kMaxRanks)];
    // the one point of a frame where the host waits for the device (a flag in mapped memory, no copy, no stream sync)
    if (int rc = slam_exchange_plan_host(e, G, plan)) return rc;
    if (plan[0] & 2) {   // the same verdict on every rank: all refuse the frame together
        snprintf(e->err, sizeof e->err, "exchange might exceed recv_capacity %d on some rank", pf->recv_cap);
        return SLAM_ERR_CAPACITY;
    }
    pf->rows_received = 0;
    if (!(plan[0] & 1) && G > 1) return SLAM_OK;   // every run boundary coincides with a rank boundary: all ranks skip
    const int32_t *scnt = plan + 1, *rcnt = plan + 1 + G;
    int64_t stot = 0, rtot = 0, sfl[kMaxRanks], rfl[kMaxRanks];
    const int64_t rec = 3 + 5 * (int64_t)L;
    for (int q = 0; q < G; ++q) {
        stot += scnt[q];
        rtot += rcnt[q];
        sfl[q] = rec * scnt[q];
        rfl[q] = rec * rcnt[q];
    }
    if (rtot > pf->recv_cap) return SLAM_ERR_CAPACITY;   // cannot happen: bit 1 above bounds it
    pf->rows_received = (int)rtot;
    if (int rc = grow(pf, &pf->sbuf, &pf->sbuf_floats, (size_t)(rec * stot))) return rc;
    if (int rc = grow(pf, &pf->rbuf, &pf->rbuf_floats, (size_t)(rec * rtot))) return rc;
    const bool split = L && pf->split, spages = split && pf->paged;   // (split pages: the means on pages of two planes, the classes as on split)
    const PageGeom geom = spages ? split_geom(pf) : PageGeom();
    const float* mp = L ? (spages ? split_pool(pf) : pf->paged ? pf->pool : split ? pf->mean[pf->sp_cur] : pf->map[pf->map_cur]) : nullptr;
    if (stot)
        if (int rc = slam_migrate_pack_paged(e, n, pf->rank, G, plan, pf->pose[pf->cur], n, mp, (split ? 2 : 5) * (int64_t)pf->Lp, pf->Lp, L,
                                             pf->sbuf, pf->paged ? pf->pt[pf->pt_cur] : nullptr, pf->nb, split ? pf->cov : nullptr,
                                             split ? pf->cls[pf->sp_cur] : nullptr, spages ? &geom : nullptr))
            return rc;
    if (int rc = comm_all_to_all_f32(pf->comm, pf->sbuf, sfl, pf->rbuf, rfl)) return rc;
    if (rtot && split) {
        // every received row becomes a class of its own (it brings its covariances along); its number comes from the free list
        // of classes on the device, made anew from the stamps when its guaranteed length — a rank's staging rows: at most n
        // classes are in use — is used up (SLAM_SPLIT_CLASS_ROOM: tests make the lists short)
        const char* room_env = getenv("SLAM_SPLIT_CLASS_ROOM");
        const int64_t room = room_env && atoi(room_env) > 0 && atoi(room_env) < pf->recv_cap ? atoi(room_env) : pf->recv_cap;
        const ProfScope prof(e, SLAM_PROF_UNPACK);
        if (pf->cls_cursor + rtot > room) {
            SLAM_HIP_TRY(e, launch_class_free_list(e->stream, pf->cstamp, pf->cap, pf->cstamp_now, pf->cls_free, pf->cls_fs));
            pf->cls_cursor = 0;
        }
        if (spages) {   // the means onto fresh pages (a new free list first if the old one runs short), table rows n .. n + rtot - 1
            int32_t* pstate = pf->page_scratch;
            SLAM_HIP_TRY(e, launch_pool_reserve(e->stream, pstate, rtot * pf->nb));
            SLAM_HIP_TRY(e, launch_free_list(e->stream, pf->stamp, pf->npages, pf->stamp_now, pf->freelist, pstate,
                                             reinterpret_cast<int32_t*>(pf->d_hres) + 20));
            SLAM_HIP_TRY(e, launch_migrate_unpack_split_pages(e->stream, pf->rbuf, (int)rtot, n, pf->pose_stage, pf->cap, split_pool(pf), geom,
                                                              pf->pt[pf->pt_cur], pf->nb, L, pf->freelist, pstate, pf->stamp, pf->stamp_now,
                                                              pf->cov, pf->covx, pf->cls[pf->sp_cur], pf->Lp, pf->cfg.meas_var, pf->cls_free,
                                                              (int)pf->cls_cursor, pf->cstamp, pf->cstamp_now, pf->live[pf->live_cur],
                                                              pf->cov_cnt + pf->cov_phase));
        } else
            SLAM_HIP_TRY(e, launch_migrate_unpack_split(e->stream, pf->rbuf, (int)rtot, n, pf->pose_stage, pf->cap, pf->mean[pf->sp_cur], pf->cov,
                                                        pf->covx, pf->cls[pf->sp_cur], pf->Lp, L, pf->cfg.meas_var, pf->cls_free,
                                                        (int)pf->cls_cursor, pf->cstamp, pf->cstamp_now, pf->live[pf->live_cur],
                                                        pf->cov_cnt + pf->cov_phase));
        pf->cls_cursor += rtot;
        pf->cls_appended += (uint32_t)rtot;
    } else if (rtot && pf->paged) {
        // fresh pages for the received rows (a new free list first if the old one runs short), table rows n .. n + rtot - 1
        int32_t* pstate = pf->page_scratch;
        const ProfScope prof(e, SLAM_PROF_UNPACK);
        SLAM_HIP_TRY(e, launch_pool_reserve(e->stream, pstate, rtot * pf->nb));
        SLAM_HIP_TRY(e, launch_free_list(e->stream, pf->stamp, pf->npages, pf->stamp_now, pf->freelist, pstate,
                                         reinterpret_cast<int32_t*>(pf->d_hres) + 20));
        SLAM_HIP_TRY(e, launch_migrate_unpack_paged(e->stream, pf->rbuf, (int)rtot, n, pf->pose_stage, pf->cap, pf->pool,
                                                    pf->pt[pf->pt_cur], pf->nb, L, pf->freelist, pstate, pf->stamp,
                                                    pf->stamp_now));
    } else if (rtot) {
        if (int rc = slam_migrate_unpack_dev(e, pf->rbuf, G, rcnt, n, pf->pose_stage, pf->cap,
                                             L ? pf->map[pf->map_cur] : nullptr, 5 * (int64_t)pf->Lp, pf->Lp, L))
            return rc;
    }
    return SLAM_OK;
}

// Every slam_pf_* call on a sharded session is collective: a rank that fails one alone (an error of its own, not a verdict
// every rank reaches together) must not leave the others waiting inside it — it gives up for good (comm_abort), the peers get
// SLAM_ERR_COMM at once instead of after SLAM_COMM_TIMEOUT_S.
int collective_result(slam_pf* pf, int rc)
{
    // (argument checks come before anything collective and are the same on every rank of a sane host: no abort for those)
    if (rc != SLAM_OK && rc != SLAM_ERR_INVALID_ARG && rc != SLAM_ERR_NOT_READY && pf && pf->comm) (void)comm_abort(pf->comm);
    return rc;
}

int finish_exchange(slam_pf* pf)
{
    if (!pf->exchange_pending) return SLAM_OK;
    pf->exchange_pending = false;
    return migrate(pf);
}

// set_poses / set_map / reset discard the pending resample gather: nothing of it may run later
int drop_resample(slam_pf* pf)
{
    pf->has_anc = false;
    pf->exchange_pending = false;
    if (pf->gated)   // ... and no weight is carried into the next frame
        if (int rc = slam_resample_gate_set(pf->e, pf->cfg.resample_ess_frac)) return rc;
    if (pf->comm) return comm_all_gather_finish(pf->comm);
    return SLAM_OK;
}

// wait for a result kernel's sequence number in mapped host memory (bounded spin, then let the runtime tell us)
int wait_result(slam_pf* pf, uint32_t seq)
{
    slam_engine* e = pf->e;
    volatile uint32_t* h_seq = reinterpret_cast<volatile uint32_t*>(pf->h_res + 16);
    if (pf->comm) return comm_wait_flag(pf->comm, h_seq, seq);
    for (long spin = 0; spin < 400000000L; ++spin)
        if (__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) == seq) return SLAM_OK;
    SLAM_HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) != seq) return slam_engine_fail_hip(e, hipErrorUnknown, "result flag");
    return SLAM_OK;
}

int gathered_copy_out(slam_pf* pf, const float* d_src, const int32_t* idx, float* h_dst, float* d_tmp)
{
    // d_src gathered through idx on the device (idx == nullptr: as is), then copied back
    if (idx) {
        int rc = slam_gather_f32_dev(pf->e, d_src, idx, pf->n, d_tmp);
        if (rc != SLAM_OK) return rc;
        d_src = d_tmp;
    }
    int rc = slam_engine_sync(pf->e);
    if (rc != SLAM_OK) return rc;
    return hipMemcpy(h_dst, d_src, sizeof(float) * (size_t)pf->n, hipMemcpyDeviceToHost) == hipSuccess ? SLAM_OK
                                                                                                      : SLAM_ERR_HIP;
}

// ---- the landmark maps live in ONE allocation `store` of 2 x cap x 5 x Lp floats, seen either as two row buffers
// (map[0] = the first half, map[1] = the second) or as a pool of 2 x cap x nb pages (the same bytes: a page is 5 x 32 floats,
// a row nb of them).  The page tables, free list and stamps exist only for sessions that may be on pages.
bool alloc_store(slam_pf* pf)
{
    const size_t half = 5 * (size_t)pf->Lp * (size_t)pf->cap;   // floats
    if (dev_alloc((void**)&pf->store, 2 * half * 4) != hipSuccess) {
        (void)hipGetLastError();
        pf->store = nullptr;
        return false;
    }
    pf->map[0] = pf->store;
    pf->map[1] = pf->store + half;
    pf->pool = pf->store;
    return true;
}

void free_page_tables(slam_pf* pf)
{
    for (void** p : { (void**)&pf->pt[0], (void**)&pf->pt[1], (void**)&pf->freelist, (void**)&pf->stamp, (void**)&pf->page_scratch,
                      (void**)&pf->votes }) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
}

bool alloc_page_tables(slam_pf* pf)
{
    const size_t P = (size_t)pf->npages, words = 4 * (size_t)pf->nb + 2 + (size_t)pool_state_words() + 4 * (size_t)pf->Lp + 2;
    const bool ok = dev_alloc((void**)&pf->freelist, P * 4) == hipSuccess && dev_alloc((void**)&pf->stamp, P * 4) == hipSuccess &&
                    hipMemset(pf->stamp, 0, P * 4) == hipSuccess && dev_alloc((void**)&pf->page_scratch, words * 4) == hipSuccess &&
                    hipMemset(pf->page_scratch, 0, words * 4) == hipSuccess &&
                    dev_alloc((void**)&pf->pt[0], (size_t)pf->cap * pf->nb * 4) == hipSuccess &&
                    dev_alloc((void**)&pf->pt[1], (size_t)pf->cap * pf->nb * 4) == hipSuccess &&
                    dev_alloc((void**)&pf->votes, 2 * 4) == hipSuccess && hipMemset(pf->votes, 0, 2 * 4) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        free_page_tables(pf);
    }
    return ok;
}

// rows -> pages while the session runs: ONE kernel, stream-ordered, no allocation.  The current rows sit in one half of the
// store (before the pending gather); they are written as pages into the OTHER half — page (other half's first page) + r * nb
// + b behind identity tables, so the pending gather index means the same thing afterwards — and the half they came from
// becomes free pages.
// Sharded: when the exchange of the last frame has already been completed (a map getter between two frames does that), the
// pending gather index also names rows of the staging tail, n .. n + rows_received - 1: they move with the rest.
int rows_to_convert(const slam_pf* pf)
{
    return pf->n + (pf->comm && pf->has_anc && !pf->exchange_pending ? pf->rows_received : 0);
}

int convert_to_pages(slam_pf* pf)
{
    slam_engine* e = pf->e;
    const int mc = pf->map_cur;
    const int page_base = (1 - mc) * pf->cap * pf->nb;
    pf->pt_cur = 0;
    SLAM_HIP_TRY(e, launch_pages_from_rows(e->stream, pf->map[mc], 5 * (int64_t)pf->Lp, pf->Lp, pf->L, pf->nb, rows_to_convert(pf),
                                           pf->pool, pf->pt[0], pf->freelist, pf->npages, pf->page_scratch, page_base));
    pf->paged = true;
    pf->conversions++;
    return SLAM_OK;
}

// pages -> rows: the pages lie anywhere in the store, a row buffer is one contiguous half of it, so the rows are put
// together in a scratch buffer first (row r = the pages table row r names, again before the pending gather) and copied
// into the first half.  While it runs this takes half as much memory again; when that is not to be had the session stays
// on pages.
int convert_to_rows(slam_pf* pf)
{
    slam_engine* e = pf->e;
    const int nrows = rows_to_convert(pf);
    const size_t used = 5 * (size_t)pf->Lp * (size_t)nrows;
    if (pf->conv_floats < used) {   // grow-only scratch (an alloc + free per move would serialise the frame every time)
        if (pf->conv_tmp) {
            if (pf->comm) {
                if (int rc = comm_wait_stream(pf->comm)) return rc;
            } else {
                SLAM_HIP_TRY(e, hipStreamSynchronize(e->stream));
            }
            (void)hipFree(pf->conv_tmp);
        }
        pf->conv_tmp = nullptr;
        pf->conv_floats = 0;
        if (hipMalloc((void**)&pf->conv_tmp, used * 4) != hipSuccess) {
            (void)hipGetLastError();
            pf->conv_tmp = nullptr;
            pf->auto_stuck = true;
            return SLAM_OK;
        }
        pf->conv_floats = used;
    }
    // stream-ordered: pages -> scratch rows -> the first half of the store (the scratch is read before anything else writes it)
    SLAM_HIP_TRY(e, launch_rows_from_pages(e->stream, pf->pool, pf->pt[pf->pt_cur], pf->nb, nullptr, nrows, pf->conv_tmp,
                                           5 * (int64_t)pf->Lp, pf->Lp, pf->L));
    SLAM_HIP_TRY(e, hipMemcpyAsync(pf->map[0], pf->conv_tmp, used * 4, hipMemcpyDeviceToDevice, e->stream));
    pf->map_cur = 0;
    pf->paged = false;
    pf->conversions++;
    return SLAM_OK;
}

// ---- split layout: tables, placement in the store, moves
void free_split_tables(slam_pf* pf)
{
    for (void** p : { (void**)&pf->cls[0], (void**)&pf->cls[1], (void**)&pf->live[0], (void**)&pf->live[1], (void**)&pf->cov_cnt,
                      (void**)&pf->cstamp, &pf->split_scratch, (void**)&pf->cls_free, (void**)&pf->cls_fs }) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
}

bool alloc_split_tables(slam_pf* pf)
{
    const size_t cap = (size_t)pf->cap;
    bool ok = true;
    for (int b = 0; b < 2; ++b)
        ok = ok && dev_alloc((void**)&pf->cls[b], cap * 4) == hipSuccess && dev_alloc((void**)&pf->live[b], cap * 4) == hipSuccess;
    ok = ok && dev_alloc((void**)&pf->cov_cnt, 16) == hipSuccess && hipMemset(pf->cov_cnt, 0, 16) == hipSuccess &&
         dev_alloc((void**)&pf->cstamp, cap * 4) == hipSuccess && hipMemset(pf->cstamp, 0, cap * 4) == hipSuccess &&
         dev_alloc(&pf->split_scratch, split_scratch_words(pf->cap) * 4) == hipSuccess &&
         dev_alloc((void**)&pf->cls_fs, 8) == hipSuccess && hipMemset(pf->cls_fs, 0, 8) == hipSuccess &&
         (!pf->comm || dev_alloc((void**)&pf->cls_free, cap * 4) == hipSuccess);
    if (!ok) {
        (void)hipGetLastError();
        free_split_tables(pf);
    }
    return ok;
}

// mean[0] and cov take the half `base` of the store (2 + 3 of its 5 units), mean[1] and covx (2 + 2) the other half
void place_split(slam_pf* pf, int base)
{
    const size_t half = 5 * (size_t)pf->Lp * (size_t)pf->cap, unit = (size_t)pf->Lp * (size_t)pf->cap;
    pf->sp_base = base;
    pf->mean[0] = pf->store + (size_t)base * half;
    pf->cov = pf->mean[0] + 2 * unit;
    pf->mean[1] = pf->store + (size_t)(1 - base) * half;
    pf->covx = pf->mean[1] + 2 * unit;
}

int32_t* split_h_live(slam_pf* pf) { return reinterpret_cast<int32_t*>(pf->d_hres) + 22; }   // {classes in use, epoch}

// ---- SPLIT PAGES (paged && split): the means on copy-on-write pages of two planes (256 bytes), the covariances per class as
// on the split layout.  The pages live in the session's two mean buffers (2 x cap x nb pages, exactly their size), which are
// not neighbours in the store: pages below cap x nb in the buffer at the lower address, the others in the other one.
float* split_pool(const slam_pf* pf) { return pf->mean[0] < pf->mean[1] ? pf->mean[0] : pf->mean[1]; }
PageGeom split_geom(const slam_pf* pf)
{
    PageGeom g;
    g.planes = 2;
    g.half_pages = (int64_t)pf->cap * pf->nb;
    const float *lo = split_pool(pf), *hi = pf->mean[0] < pf->mean[1] ? pf->mean[1] : pf->mean[0];
    g.gap = (hi - lo) - g.half_pages * 2 * kPageLandmarks;
    return g;
}
bool split_pages(const slam_pf* pf) { return pf->paged && pf->split; }

// a new set of classes is about to be made (set_map, reset, rows -> split): lists and counters start afresh
void split_new_epoch(slam_pf* pf)
{
    pf->cls_epoch++;
    pf->cstamp_now++;
    pf->live_cur = 0;
    pf->cov_phase = 0;
    pf->cls_appended = 0;
    pf->cls_cursor = (int64_t)1 << 40;   // no list of free class numbers yet: the first arrivals make one
}

// nrows rows (as given, any strides) -> means + classes + class rows in the buffers of the current placement
int split_from_rows(slam_pf* pf, const float* d_rows, int64_t row_stride, int plane_stride, int nrows)
{
    slam_engine* e = pf->e;
    split_new_epoch(pf);
    SLAM_HIP_TRY(e, launch_split_from_rows(e->stream, d_rows, row_stride, plane_stride, pf->L, nrows, pf->Lp, pf->mean[pf->sp_cur], pf->cov,
                                           pf->covx, pf->cfg.meas_var, pf->cls[pf->sp_cur], pf->live[0], pf->cov_cnt, 0, pf->cstamp,
                                           pf->cstamp_now, split_h_live(pf), pf->cls_epoch, pf->split_scratch));
    return SLAM_OK;
}

// rows -> split while the session runs: stream-ordered, no allocation.  The rows sit in one half of the store; the means and
// the class rows go into the OTHER half, and the half the rows came from becomes the second mean buffer.
int convert_rows_to_split(slam_pf* pf)
{
    const int mc = pf->map_cur;
    place_split(pf, 1 - mc);
    pf->sp_cur = 0;
    if (int rc = split_from_rows(pf, pf->map[mc], 5 * (int64_t)pf->Lp, pf->Lp, rows_to_convert(pf))) return rc;
    pf->split = true;
    pf->conversions++;
    return SLAM_OK;
}

// split -> split pages: the means of the current buffer become pages in the OTHER mean buffer (identity tables, shifted),
// the buffer they came from becomes free pages; classes and covariances stay where they are.  One stream-ordered launch.
int convert_split_to_split_pages(slam_pf* pf)
{
    slam_engine* e = pf->e;
    const float* src = pf->mean[pf->sp_cur];
    const float* dst = pf->mean[1 - pf->sp_cur];
    const int page_base = dst == split_pool(pf) ? 0 : pf->cap * pf->nb;
    pf->pt_cur = 0;
    // (sharded: rows_to_convert takes the staging tail along when the exchange of the last frame has been completed already)
    SLAM_HIP_TRY(e, launch_pages_from_rows(e->stream, src, 2 * (int64_t)pf->Lp, pf->Lp, pf->L, pf->nb, rows_to_convert(pf), split_pool(pf),
                                           pf->pt[0], pf->freelist, pf->npages, pf->page_scratch, page_base, split_geom(pf)));
    pf->paged = true;
    pf->conversions++;
    return SLAM_OK;
}

// split pages -> split: the pages lie anywhere in the two mean buffers, so the rows are put together in the scratch buffer of
// convert_to_rows first and copied into mean[0]
int convert_split_pages_to_split(slam_pf* pf)
{
    slam_engine* e = pf->e;
    const int nrows = rows_to_convert(pf);
    const size_t used = 2 * (size_t)pf->Lp * (size_t)nrows;
    if (pf->conv_floats < used) {
        if (pf->conv_tmp) {
            if (pf->comm) {
                if (int rc = comm_wait_stream(pf->comm)) return rc;
            } else {
                SLAM_HIP_TRY(e, hipStreamSynchronize(e->stream));
            }
            (void)hipFree(pf->conv_tmp);
        }
        pf->conv_tmp = nullptr;
        pf->conv_floats = 0;
        if (hipMalloc((void**)&pf->conv_tmp, used * 4) != hipSuccess) {
            (void)hipGetLastError();
            pf->conv_tmp = nullptr;
            pf->auto_stuck = true;
            return SLAM_OK;
        }
        pf->conv_floats = used;
    }
    SLAM_HIP_TRY(e, launch_rows_from_pages(e->stream, split_pool(pf), pf->pt[pf->pt_cur], pf->nb, nullptr, nrows, pf->conv_tmp,
                                           2 * (int64_t)pf->Lp, pf->Lp, pf->L, split_geom(pf)));
    SLAM_HIP_TRY(e, hipMemcpyAsync(pf->mean[0], pf->conv_tmp, used * 4, hipMemcpyDeviceToDevice, e->stream));
    if (pf->sp_cur == 1) SLAM_HIP_TRY(e, hipMemcpyAsync(pf->cls[0], pf->cls[1], (size_t)nrows * 4, hipMemcpyDeviceToDevice, e->stream));
    pf->sp_cur = 0;
    pf->paged = false;
    pf->conversions++;
    return SLAM_OK;
}

// SLAM_MAP_AUTO, at the start of a frame: look at the counts that have arrived since the last look (no waiting) and move
// when the last three agree.  Pages pay when a frame observes at most two sevenths of the landmarks (a resampling frame on
// rows rewrites every row in full); rows / split maps pay when it observes more than three eighths (most pages are touched
// anyway and the row kernels are faster at that; the measured change-over is at 0.28-0.33).  Results do not depend on the layout, so the ranks of a sharded session may decide apart.
int auto_layout(slam_pf* pf)
{
    if (pf->layout_cfg != SLAM_MAP_AUTO || pf->L == 0 || pf->auto_stuck) return SLAM_OK;
    const int32_t* h = reinterpret_cast<const int32_t*>(pf->h_res) + 24;   // {observed, L, seq, votes_pages, votes_rows}
    const volatile uint32_t* h_seq = reinterpret_cast<const volatile uint32_t*>(h) + 2;
    // The first frames of a session WAIT for the sample of the frame before (it is taken early in that frame: the wait is
    // about one motion + score launch), so that a session settles on its layout within its first four frames however far
    // the host runs ahead of the device; later looks never wait.
    if (pf->frame <= 3 && pf->obs_seq_issued != pf->obs_seq_seen) {
        if (pf->comm) {
            if (int rc = comm_wait_flag(pf->comm, h_seq, pf->obs_seq_issued)) return rc;
        } else {
            for (long spin = 0; spin < 400000000L && __atomic_load_n(h_seq, __ATOMIC_ACQUIRE) != pf->obs_seq_issued; ++spin) {}
        }
    }
    const uint32_t seq = __atomic_load_n(h_seq, __ATOMIC_ACQUIRE);
    if (seq == pf->obs_seq_seen) return SLAM_OK;
    pf->obs_seq_seen = seq;
    pf->votes_pages = h[3];   // samples in a row (counted on the device, so none is missed however far the host runs ahead)
    pf->votes_rows = h[4];
    if (!pf->paged && pf->votes_pages >= 3) {
        if (pf->split) return convert_split_to_split_pages(pf);   // the means go onto pages, the classes stay
        return convert_to_pages(pf);
    }
    if (pf->paged && pf->votes_rows >= 3) {
        if (pf->split) return convert_split_pages_to_split(pf);
        if (int rc = convert_to_rows(pf)) return rc;
        if (!pf->paged && pf->dense_split) return convert_rows_to_split(pf);
    }
    return SLAM_OK;
}

int create_common(slam_engine* e, const slam_pf_config* cfg, slam_comm* comm, int recv_capacity, slam_pf** out)
{
    if (!e || !cfg || !out || cfg->n_particles <= 0 || cfg->n_landmarks < 0 || !(cfg->meas_var > 0.0f) ||
        cfg->map_layout < SLAM_MAP_AUTO || cfg->map_layout > SLAM_MAP_SPLIT_PAGES)
        return SLAM_ERR_INVALID_ARG;

    *out = nullptr;
    if (e->live_sessions > 0) {   // the stages keep per-population state in the engine (gate, carried weights, exchange plan)
        snprintf(e->err, sizeof e->err, "this engine already runs a particle-filter session: one session per engine");
        return SLAM_ERR_NOT_READY;
    }
    if (comm && comm_engine(comm) != e) return SLAM_ERR_INVALID_ARG;
    if (int rc = slam_engine_sync(e)) return rc;   // also selects the engine's device
    slam_pf* pf = new (std::nothrow) slam_pf();
    if (!pf) return SLAM_ERR_HIP;
    pf->e = e;
    pf->cfg = *cfg;
    pf->n = cfg->n_particles;
    pf->L = cfg->n_landmarks;
    pf->Lp = (pf->L + 31) / 32 * 32;
    pf->comm = comm;
    if (comm) {
        pf->rank = comm_rank(comm);
        pf->world = comm_world(comm);
        pf->recv_cap = recv_capacity > 0 && recv_capacity < pf->n ? recv_capacity : pf->n;
    }
    pf->cap = pf->n + pf->recv_cap;
    pf->n_total = (int64_t)pf->n * pf->world;
    if (3 * pf->n_total > 0x7fffffff) {   // int32 ancestor indices into the all-gathered pose array
        delete pf;
        return SLAM_ERR_CAPACITY;
    }
    const size_t n = (size_t)pf->n, L = (size_t)pf->L, cap = (size_t)pf->cap, G = (size_t)pf->world;
    bool ok = true;
    // slam_pf_paged_set(e, 1) turns an AUTO request into PAGES (sessions made while it is set stay on pages)
    pf->layout_cfg = cfg->map_layout == SLAM_MAP_AUTO && e->pf_paged ? (int)SLAM_MAP_PAGES : cfg->map_layout;
    pf->paged = pf->L > 0 && (pf->layout_cfg == SLAM_MAP_PAGES || pf->layout_cfg == SLAM_MAP_SPLIT_PAGES);
    pf->nb = pf->Lp / kPageLandmarks;
    {
        const int64_t np = 2 * (int64_t)pf->cap * pf->nb;   // table rows (with the staging tail) never name more than half
        // page numbers are int32, and free_list_kernel's last tile may look 8191 past the end
        if (np > 0x7fffffff - 8192) {
            if (pf->paged) {
                delete pf;
                return SLAM_ERR_CAPACITY;
            }
            pf->auto_stuck = true;   // too many pages for this population: AUTO stays on rows
        }
        pf->npages = np > 0x7fffffff - 8192 ? 0 : (int)np;
        if (pf->nb < 2) pf->auto_stuck = true;   // one page per particle: nothing to gain from pages
    }
    pf->gated = cfg->resample_ess_frac > 0.0f && cfg->resample_ess_frac < 1.0f;
    if (L) ok = alloc_store(pf);
    // the split layout: asked for, or what AUTO keeps a session on while its frames observe most landmarks.  (Round 4 first kept
    // ESS-gated sessions on rows, whose update of a frame that keeps its population runs in place on the observed landmarks; the
    // split update of such a frame goes through the identity index out of place and is faster all the same: 65 536 x 500,
    // every landmark observed, 0.113 against 0.180 ms per frame; 32 observed on split pages 0.088 against 0.142-0.169.)
    const bool want_split = pf->layout_cfg == SLAM_MAP_SPLIT || pf->layout_cfg == SLAM_MAP_SPLIT_PAGES;
    if (L && ok && (want_split || pf->layout_cfg == SLAM_MAP_AUTO)) {
        const bool have = alloc_split_tables(pf);
        if (!have && want_split) ok = false;
        pf->dense_split = have && pf->layout_cfg == SLAM_MAP_AUTO;
        pf->split = have;
        if (have) place_split(pf, 0);
    }
    if (L && ok && (pf->paged || (pf->layout_cfg == SLAM_MAP_AUTO && !pf->auto_stuck))) {
        ok = alloc_page_tables(pf);
        if (!ok && !pf->paged) {   // AUTO can live without them: it stays on rows
            pf->auto_stuck = true;
            ok = true;
        }
    }
    for (int b = 0; b < 2; ++b) {
        ok = ok && dev_alloc((void**)&pf->pose[b], 3 * n * 4) == hipSuccess;
        ok = ok && dev_alloc((void**)&pf->anc[b], n * 4) == hipSuccess;
        if (comm) ok = ok && dev_alloc((void**)&pf->pose_idx[b], n * 4) == hipSuccess;
    }
    ok = ok && dev_alloc((void**)&pf->score, n * 4) == hipSuccess && dev_alloc((void**)&pf->logw, n * 4) == hipSuccess &&
         dev_alloc((void**)&pf->count, n * 4) == hipSuccess && dev_alloc((void**)&pf->first, n * 4) == hipSuccess &&
         dev_alloc((void**)&pf->res_dev, 64) == hipSuccess && dev_alloc((void**)&pf->sums_acc, 64) == hipSuccess &&
         dev_alloc(&pf->res_all, 64 * G) == hipSuccess &&
         hipMemset(pf->sums_acc, 0, 64) == hipSuccess;
    // results come back through the engine's mapped buffer (one session per engine; see engine_internal.h for why the
    // session does not allocate its own); the engine was drained above, so nothing of an earlier session writes to it any more
    pf->h_res = static_cast<decltype(pf->h_res)>(e->h_pf_res);
    pf->d_hres = static_cast<decltype(pf->d_hres)>(e->d_hpf_res);
    if (ok) memset(e->h_pf_res, 0, 128);
    if (comm)
        ok = ok && dev_alloc((void**)&pf->pose_all, 3 * n * G * 4) == hipSuccess &&
             dev_alloc((void**)&pf->pose_stage, 3 * cap * 4) == hipSuccess &&
             dev_alloc((void**)&pf->first_all, n * G * 4) == hipSuccess && dev_alloc((void**)&pf->d_max, 4) == hipSuccess &&
             dev_alloc((void**)&pf->d_sum, 3 * 8) == hipSuccess && dev_alloc((void**)&pf->totals, 3 * 8 * G) == hipSuccess &&
             dev_alloc((void**)&pf->d_plan, 4 * SLAM_PLAN_WORDS(kMaxRanks)) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        slam_pf_destroy(pf);
        return SLAM_ERR_HIP;
    }
    if (comm)
        if (int rc = slam_exchange_set_capacity(e, pf->recv_cap)) {
            slam_pf_destroy(pf);
            return rc;
        }
    if (int rc = slam_resample_gate_set(e, pf->gated ? cfg->resample_ess_frac : 0.0f)) {
        slam_pf_destroy(pf);
        return rc;
    }
    const float origin[3] = { 0, 0, 0 };
    if (int rc = slam_pf_reset(pf, origin)) {   // a failed reset must not hand back a live object with an error code
        slam_pf_destroy(pf);
        return rc;
    }
    e->live_sessions++;
    pf->counted = true;
    *out = pf;
    return SLAM_OK;
}

}  // namespace

extern "C" {

int slam_pf_create(slam_engine* e, const slam_pf_config* cfg, slam_pf** out)
{
    return create_common(e, cfg, nullptr, 0, out);
}

int slam_pf_create_sharded(slam_engine* e, const slam_pf_config* cfg, slam_comm* comm, int recv_capacity, slam_pf** out)
{
    if (!comm) return SLAM_ERR_INVALID_ARG;
    return create_common(e, cfg, comm, recv_capacity, out);
}

int slam_pf_destroy(slam_pf* pf)
{
    if (!pf) return SLAM_OK;
    (void)slam_engine_sync(pf->e);
    if (pf->counted) pf->e->live_sessions--;
    if (pf->comm) {
        (void)comm_all_gather_finish(pf->comm);
        (void)slam_engine_sync(pf->e);
        (void)slam_exchange_set_capacity(pf->e, 0);
    }
    if (pf->gated) (void)slam_resample_gate_set(pf->e, 0.0f);
    for (int b = 0; b < 2; ++b) {
        (void)hipFree(pf->pose[b]);
        (void)hipFree(pf->anc[b]);
        (void)hipFree(pf->pose_idx[b]);
    }
    if (pf->store) (void)hipFree(pf->store);
    free_page_tables(pf);
    free_split_tables(pf);
    for (void* p : { (void*)pf->score, (void*)pf->logw, (void*)pf->count, (void*)pf->first, (void*)pf->pose_all, (void*)pf->pose_stage, (void*)pf->first_all,
                     (void*)pf->d_max, (void*)pf->d_sum, (void*)pf->totals, (void*)pf->d_plan, (void*)pf->sbuf,
                     (void*)pf->rbuf, (void*)pf->res_dev, (void*)pf->sums_acc, pf->res_all })
        (void)hipFree(p);
    (void)hipFree(pf->sel);
    (void)hipFree(pf->conv_tmp);
    delete pf;   // h_res is the engine's
    return SLAM_OK;
}

int slam_pf_reset(slam_pf* pf, const float pose[3])
{
    if (!pf || !pose) return SLAM_ERR_INVALID_ARG;
    const size_t n = (size_t)pf->n;
    std::vector<float> h(3 * n);
    for (int k = 0; k < 3; ++k)
        for (size_t i = 0; i < n; ++i) h[k * n + i] = pose[k];
    if (int rc = drop_resample(pf)) return rc;
    if (int rc = slam_engine_sync(pf->e)) return rc;
    if (hipMemcpy(pf->pose[pf->cur], h.data(), 3 * n * 4, hipMemcpyHostToDevice) != hipSuccess) return SLAM_ERR_HIP;
    if (pf->split) {   // every landmark of every particle "not seen yet": one class
        split_new_epoch(pf);
        SLAM_HIP_TRY(pf->e, launch_split_reset(pf->e->stream, pf->mean[pf->sp_cur], pf->cov, pf->covx, pf->cls[pf->sp_cur], pf->Lp, pf->n,
                                               pf->live[0], pf->cov_cnt, 0, pf->cstamp, pf->cstamp_now, split_h_live(pf), pf->cls_epoch));
        if (pf->paged)   // split pages: every particle names ONE shared page of zero means, the rest of the pool is free
            SLAM_HIP_TRY(pf->e, launch_pages_reset(pf->e->stream, split_pool(pf), pf->pt[pf->pt_cur], (int64_t)pf->n * pf->nb, pf->freelist,
                                                   pf->npages, pf->page_scratch, split_geom(pf)));
        if (int rc = slam_engine_sync(pf->e)) return rc;
    } else if (pf->paged) {   // every particle names ONE shared page of landmarks not seen yet
        SLAM_HIP_TRY(pf->e, launch_pages_reset(pf->e->stream, pf->pool, pf->pt[pf->pt_cur], (int64_t)pf->n * pf->nb, pf->freelist,
                                               pf->npages, pf->page_scratch));
        if (int rc = slam_engine_sync(pf->e)) return rc;
    } else if (pf->L) {   // P_xx = -1: "not seen yet"
        const size_t Lp = (size_t)pf->Lp;
        std::vector<float> row(5 * Lp, 0.0f);
        for (size_t l = 0; l < Lp; ++l) row[2 * Lp + l] = -1.0f;
        // one row on the host, replicated over the particles in chunks (a 52 GB map does not pass through host memory)
        const size_t chunk = n < 4096 ? n : 4096;
        std::vector<float> m(5 * Lp * chunk);
        for (size_t i = 0; i < chunk; ++i) memcpy(&m[i * 5 * Lp], row.data(), 5 * Lp * 4);
        for (size_t i0 = 0; i0 < n; i0 += chunk) {
            const size_t k = n - i0 < chunk ? n - i0 : chunk;
            if (hipMemcpy(pf->map[pf->map_cur] + i0 * 5 * Lp, m.data(), k * 5 * Lp * 4, hipMemcpyHostToDevice) != hipSuccess)
                return SLAM_ERR_HIP;
        }
    }
    pf->frame = 0;
    return SLAM_OK;
}

int slam_pf_set_poses_host(slam_pf* pf, const float* x, const float* y, const float* theta)
{
    if (!pf || !x || !y || !theta) return SLAM_ERR_INVALID_ARG;
    const size_t n = (size_t)pf->n;
    if (int rc = drop_resample(pf)) return rc;
    if (int rc = slam_engine_sync(pf->e)) return rc;
    float* d = pf->pose[pf->cur];
    if (hipMemcpy(d, x, n * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + n, y, n * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + 2 * n, theta, n * 4, hipMemcpyHostToDevice) != hipSuccess)
        return SLAM_ERR_HIP;
    return SLAM_OK;
}

int slam_pf_set_map_host(slam_pf* pf, const float* rows)
{
    if (!pf || !rows || !pf->L) return SLAM_ERR_INVALID_ARG;
    if (int rc = slam_engine_sync(pf->e)) return rc;
    if (pf->has_anc) return SLAM_ERR_NOT_READY;   // set poses / reset first: a gather is pending
    // host [n][5][L] -> device [n][5][Lp]: 5n planes of L floats each
    const bool indirect = pf->paged || pf->split;   // no rows to copy into: a scratch copy goes through slam_pf_set_map_dev
    float* dense = indirect ? nullptr : pf->map[pf->map_cur];
    if (indirect && hipMalloc((void**)&dense, 5 * (size_t)pf->Lp * pf->n * 4) != hipSuccess) return SLAM_ERR_HIP;
    int rc = SLAM_OK;
    if (hipMemcpy2D(dense, (size_t)pf->Lp * 4, rows, (size_t)pf->L * 4, (size_t)pf->L * 4, 5 * (size_t)pf->n,
                    hipMemcpyHostToDevice) != hipSuccess)
        rc = SLAM_ERR_HIP;
    if (indirect) {
        if (rc == SLAM_OK) rc = slam_pf_set_map_dev(pf, dense, 5 * (int64_t)pf->Lp, pf->Lp);
        if (rc == SLAM_OK) rc = slam_engine_sync(pf->e);
        (void)hipFree(dense);
    }
    return rc;
}

int slam_pf_set_map_dev(slam_pf* pf, const float* d_rows, int64_t row_stride, int plane_stride)
{
    if (!pf || !d_rows || !pf->L || plane_stride < pf->L || row_stride < 5 * (int64_t)plane_stride) return SLAM_ERR_INVALID_ARG;
    if (pf->has_anc) return SLAM_ERR_NOT_READY;   // set poses / reset first: a gather is pending
    slam_engine* e = pf->e;
    SLAM_HIP_TRY(e, hipSetDevice(e->device));
    if (pf->split && pf->paged) {   // split pages: classes and mean rows first, then the means onto pages
        pf->paged = false;
        pf->sp_cur = 0;
        if (int rc = split_from_rows(pf, d_rows, row_stride, plane_stride, pf->n)) return rc;
        if (int rc = convert_split_to_split_pages(pf)) return rc;
        pf->conversions--;
        return SLAM_OK;
    }
    if (pf->split) return split_from_rows(pf, d_rows, row_stride, plane_stride, pf->n);
    if (pf->paged) {
        SLAM_HIP_TRY(e, launch_pages_from_rows(e->stream, d_rows, row_stride, plane_stride, pf->L, pf->nb, pf->n, pf->pool,
                                               pf->pt[pf->pt_cur], pf->freelist, pf->npages, pf->page_scratch));
        return SLAM_OK;
    }
    for (int pl = 0; pl < 5; ++pl)   // plane by plane: one 2-D copy each whatever the caller's row stride is
        SLAM_HIP_TRY(e, hipMemcpy2DAsync(pf->map[pf->map_cur] + (size_t)pl * pf->Lp, 5 * (size_t)pf->Lp * 4,
                                         d_rows + (size_t)pl * plane_stride, (size_t)row_stride * 4, (size_t)pf->L * 4,
                                         (size_t)pf->n, hipMemcpyDeviceToDevice, e->stream));
    return SLAM_OK;
}

int slam_pf_is_paged(const slam_pf* pf) { return pf && pf->paged ? 1 : 0; }

// The classes' update of a frame (cov_update_kernel), in place, once per class still in use; nlandmarks = 0: only the list of
// classes in use is brought up to date (a frame without observations).  The launch is as wide as the host knows the list to
// be at most: its length as of some earlier launch (mapped memory, read without waiting) plus the classes that arrived since
// (sharded sessions) — the second word is the running count of arrivals as of that launch; it is read FIRST and written
// last, so a torn pair only over-estimates; before anything of this epoch has arrived: every class there can be.
// the arguments and the width of the classes' update of this frame (and the session's bookkeeping moved on as if it had been
// launched: the caller launches it, by itself or inside the launch of the weights)
static void split_class_prepare(slam_pf* pf, int nlandmarks, CovArgs& ca, int& bound)
{
    slam_engine* e = pf->e;
    const int32_t* hw = reinterpret_cast<const int32_t*>(pf->h_res);   // words 22-23: {count, epoch}; 18-19: {mark, epoch}
    const uint64_t hm = __atomic_load_n(reinterpret_cast<const uint64_t*>(hw + 18), __ATOMIC_ACQUIRE),
                   hl = __atomic_load_n(reinterpret_cast<const uint64_t*>(hw + 22), __ATOMIC_ACQUIRE);
    const bool fresh = (uint32_t)(hl >> 32) == pf->cls_epoch && (uint32_t)hl > 0;
    const uint32_t mark = (uint32_t)(hm >> 32) == pf->cls_epoch ? (uint32_t)hm : 0u;   // (no launch of this epoch has said yet: 0)
    const int64_t upper = fresh ? (int64_t)(uint32_t)hl + (int64_t)(pf->cls_appended - mark) : (int64_t)pf->cap;
    bound = upper < pf->cap ? (int)upper : pf->cap;
    ca.cov = pf->cov;
    ca.cov_stride = 3 * (int64_t)pf->Lp;
    ca.covx = pf->covx;
    ca.covx_stride = 2 * (int64_t)pf->Lp;
    ca.plane_stride = pf->Lp;
    ca.nlandmarks = nlandmarks;
    ca.obs_zx = e->d_obs_zx;
    ca.obs_zy = e->d_obs_zy;
    ca.meas_var = pf->cfg.meas_var;
    ca.live_in = pf->live[pf->live_cur];
    ca.live_out = pf->live[1 - pf->live_cur];
    ca.cnt = pf->cov_cnt;
    ca.phase = pf->cov_phase;
    ca.cstamp = pf->cstamp;
    ca.stamp_now = pf->cstamp_now;
    ca.h_live = split_h_live(pf);
    ca.h_mark = reinterpret_cast<int32_t*>(pf->d_hres) + 18;
    ca.epoch = pf->cls_epoch;
    ca.mark = pf->cls_appended;
    pf->live_cur = 1 - pf->live_cur;
    pf->cov_phase = (pf->cov_phase + 1) % 3;
}

static int split_class_update(slam_pf* pf, int nlandmarks)
{
    CovArgs ca;
    int bound = 0;
    split_class_prepare(pf, nlandmarks, ca, bound);
    SLAM_HIP_TRY(pf->e, launch_cov_update(pf->e->stream, ca, bound, pf->e->prof_next(SLAM_PROF_PAGES)));
    return SLAM_OK;
}

// the classes' update of the frame + the weights: ONE launch (SLAM_COV_MERGE=0: two, as round 4 first had them)
static int weights_with_classes(slam_pf* pf, int nlandmarks, bool use_ekf, float* d_max)
{
    static const bool merge = !(getenv("SLAM_COV_MERGE") && atoi(getenv("SLAM_COV_MERGE")) == 0);
    slam_engine* e = pf->e;
    if (!merge) {
        if (int rc = split_class_update(pf, nlandmarks)) return rc;
        return use_ekf ? slam_logweight_ekf_dev(e, pf->score, pf->cfg.score_gain, pf->n, pf->logw, d_max)
                       : slam_logweight_dev(e, pf->score, nullptr, pf->cfg.score_gain, pf->n, pf->logw, d_max);
    }
    CovArgs ca;
    int bound = 0;
    split_class_prepare(pf, nlandmarks, ca, bound);
    return slam_logweight_cov_dev(e, pf->score, use_ekf, pf->cfg.score_gain, pf->n, pf->logw, d_max, &ca, bound);
}

static int pf_step_impl(slam_pf* pf, int slot, const float dp[3], int use_observations, bool* collective_verdict);

int slam_pf_step(slam_pf* pf, int slot, const float dp[3], int use_observations)
{
    if (!pf || !dp) return SLAM_ERR_INVALID_ARG;
    bool collective_verdict = false;
    const int rc = pf_step_impl(pf, slot, dp, use_observations, &collective_verdict);
    // A frame that fails on THIS rank alone leaves the other ranks inside (or on their way into) a collective: give up
    // for good so that they get SLAM_ERR_COMM instead of waiting for ever.  A refusal every rank reaches together (the
    // staging area might overflow: bit 1 of the plan) is not such a failure.
    if (rc != SLAM_OK && pf->comm && !collective_verdict) (void)comm_abort(pf->comm);
    return rc;
}

static int pf_step_impl(slam_pf* pf, int slot, const float dp[3], int use_observations, bool* collective_verdict)
{
    slam_engine* e = pf->e;
    slam_comm* comm = pf->comm;
    if (pf->paged && __atomic_load_n(reinterpret_cast<int32_t*>(pf->h_res) + 20, __ATOMIC_ACQUIRE) != 0) {
        snprintf(e->err, sizeof e->err, "paged maps: a free list was shorter than the pages reserved from it (pool invariant broken)");
        return SLAM_ERR_CAPACITY;
    }
    if (int rc0 = auto_layout(pf)) return rc0;   // may move the maps between rows and pages (never changes a result)
    const int n = pf->n, L = pf->L, cur = pf->cur, nxt = 1 - cur;
    const size_t sn = (size_t)n;
    // SLAM_MAP_AUTO samples the number of observed landmarks: every frame at the start and while the counts speak against
    // the current layout, every 8th frame otherwise
    const bool sample_obs = pf->layout_cfg == SLAM_MAP_AUTO && !pf->auto_stuck && L > 0 && use_observations &&
                            e->obs_nlandmarks == L &&
                            (pf->frame < 8 || (pf->frame & 7u) == 0 || (pf->paged ? pf->votes_rows : pf->votes_pages) > 0);
    int32_t* d_hobs = reinterpret_cast<int32_t*>(pf->d_hres) + 24;
    const float* src = pf->pose[cur];
    float* dst = pf->pose[nxt];
    const int32_t* anc = pf->has_anc ? pf->anc[cur] : nullptr;
    const int64_t first_id = (int64_t)pf->rank * n;
    int rc;
    // 1 + 2 + 3 in ONE launch when the frame allows it (a single-GPU session on rows that resamples every frame, landmarks
    // observed, long rows, enough particles): motion sample + scan-match score and the out-of-place landmark update side by
    // side, the scorer's gathers in the shadow of the update's row stores (slam_frame_front_dev; the same bits)
    bool fused = false;
    SplitIO sio{};
    auto make_sio = [&]() {   // the classes follow their particles through the update; it stamps the ones still in use
        sio.group_filter = 0;
        sio.map_anc = nullptr;
        sio.cov = pf->cov;
        sio.cov_stride = 3 * (int64_t)pf->Lp;
        sio.covx = pf->covx;
        sio.covx_stride = 2 * (int64_t)pf->Lp;
        sio.cls_in = pf->cls[pf->sp_cur];
        sio.cls_out = pf->cls[1 - pf->sp_cur];
        sio.cstamp = pf->cstamp;
        sio.stamp_now = pf->cstamp_now + 1;
    };
    if (pf->split) make_sio();
    // Paged maps: the frame's page list (touched pages, observation list, where the fresh pages come from) and, when that asked
    // for one, a new free list.  Neither needs anything from the motion + score launch nor the other way round: on one GPU the
    // page list goes out first and the free list travels in workgroups of the scorer's launch (free_list_body.h); a sharded
    // session issues both behind its exchange, which takes pages from the same list first.
    int32_t *pstate = pf->page_scratch, *count = pstate + pool_state_words(), *tpage = count + 1, *tindex = tpage + pf->nb,
            *tmask = tindex + pf->nb, *tbase = tmask + pf->nb, *lst = tbase + pf->nb + 1;
    // the list form (one lane per observation) whenever a list can be made; SLAM_PAGED_FORM=0 keeps the page-wide form
    static const int env_form = getenv("SLAM_PAGED_FORM") ? atoi(getenv("SLAM_PAGED_FORM")) : 1;
    const int form = env_form != 0 && L <= kObsListMaxLandmarks ? 1 : 0;
    ObsListOut lo;
    if (form && pf->paged) {
        lo.id = lst;
        lo.zx = reinterpret_cast<float*>(lst + pf->Lp);
        lo.zy = reinterpret_cast<float*>(lst + 2 * pf->Lp);
        lo.round = lst + 3 * pf->Lp;
        lo.count = lst + 4 * pf->Lp;
    }
    auto issue_page_list = [&]() -> int {
        const ProfScope prof(e, SLAM_PROF_PAGES);
        SLAM_HIP_TRY(e, launch_page_list(e->stream, e->d_obs_zx, e->d_obs_zy, L, pf->nb, tpage, tindex, tmask, tbase, count, n, pstate,
                                         sample_obs ? d_hobs : nullptr, sample_obs ? ++pf->obs_seq_issued : 0,
                                         sample_obs ? pf->votes : nullptr, reinterpret_cast<int32_t*>(pf->d_hres) + 30, lo));
        return SLAM_OK;
    };
    static const bool ride = !(getenv("SLAM_FREE_LIST_RIDER") && atoi(getenv("SLAM_FREE_LIST_RIDER")) == 0);
    bool paged_listed = false;
    FreeListRider rider;
    if (ride && !comm && pf->paged && L > 0 && use_observations && e->obs_nlandmarks == L) {
        if (int rc1 = issue_page_list()) return rc1;
        rider.stamp = pf->stamp;
        rider.npages = pf->npages;
        rider.live = pf->stamp_now;
        rider.freelist = pf->freelist;
        rider.pool_state = pstate;
        rider.h_short = reinterpret_cast<int32_t*>(pf->d_hres) + 20;
        paged_listed = true;
    }
    if (comm && pf->split && !pf->paged && pf->has_anc && !pf->gated && anc && L > 0 && use_observations && e->obs_nlandmarks == L) {
        // Sharded, split maps: the front launch scores every particle (its ancestor's pose comes out of the all-gathered poses)
        // and updates the groups of particles whose ancestors are all rows of this rank; the groups with an ancestor in the
        // staging tail follow behind the exchange (below).  Like the motion + score launch it replaces, it needs nothing from
        // the exchange and goes out before the host has looked at the plan.
        if ((rc = comm_all_gather_finish(comm)) != SLAM_OK) return rc;
        const float* pa = pf->pose_all;
        sio.group_filter = 1;
        sio.map_anc = anc;
        rc = slam_frame_front_dev(e, slot, pa, pa + sn, pa + 2 * sn, pf->pose_idx[cur], dst, dst + sn, dst + 2 * sn, n, first_id, dp,
                                  pf->cfg.sigma, pf->cfg.seed, pf->frame, pf->score, pf->count, pf->mean[pf->sp_cur],
                                  pf->mean[1 - pf->sp_cur], 2 * (int64_t)pf->Lp, pf->Lp, L, pf->cfg.meas_var, &fused, &sio);
        if (rc != SLAM_OK) return rc;
    }
    // (a gated session on rows is left out: its frames that keep their population update in place; on the split layout they
    // go through the identity index the resample stage leaves, like any other frame)
    if (!comm && !pf->paged && (!pf->gated || pf->split) && anc && L > 0 && use_observations && e->obs_nlandmarks == L) {
        if (pf->split)
            rc = slam_frame_front_dev(e, slot, src, src + sn, src + 2 * sn, anc, dst, dst + sn, dst + 2 * sn, n, first_id, dp,
                                      pf->cfg.sigma, pf->cfg.seed, pf->frame, pf->score, pf->count, pf->mean[pf->sp_cur],
                                      pf->mean[1 - pf->sp_cur], 2 * (int64_t)pf->Lp, pf->Lp, L, pf->cfg.meas_var, &fused, &sio);
        else
            rc = slam_frame_front_dev(e, slot, src, src + sn, src + 2 * sn, anc, dst, dst + sn, dst + 2 * sn, n, first_id, dp,
                                      pf->cfg.sigma, pf->cfg.seed, pf->frame, pf->score, pf->count, pf->map[pf->map_cur],
                                      pf->map[1 - pf->map_cur], 5 * (int64_t)pf->Lp, pf->Lp, L, pf->cfg.meas_var, &fused);
        if (rc != SLAM_OK) return rc;
    }
    // 1 + 2. motion (+ the fused gather of the previous resample) and scan-match score, one launch.  Sharded: the
    // ancestors' poses come out of the array of every rank's poses, so this launch needs nothing from the exchange
    // below and keeps the GPU busy while the host picks up the exchange plan.
    if (fused) {
        rc = SLAM_OK;
    } else if (comm && pf->has_anc) {
        if ((rc = comm_all_gather_finish(comm)) != SLAM_OK) return rc;
        const float* pa = pf->pose_all;
        rc = slam_motion_score_dev(e, slot, pa, pa + sn, pa + 2 * sn, pf->pose_idx[cur], dst, dst + sn, dst + 2 * sn, n,
                                   first_id, dp, pf->cfg.sigma, pf->cfg.seed, pf->frame, pf->score, pf->count);
    } else {
        bool rode = false;
        rc = slam_motion_score_rider_dev(e, slot, src, src + sn, src + 2 * sn, anc, dst, dst + sn, dst + 2 * sn, n, first_id, dp,
                                         pf->cfg.sigma, pf->cfg.seed, pf->frame, pf->score, pf->count, paged_listed ? &rider : nullptr,
                                         &rode);
        if (rc == SLAM_OK && paged_listed && !rode) {   // (a small population: its scorer has no room for a rider)
            const ProfScope prof(e, SLAM_PROF_PAGES);
            SLAM_HIP_TRY(e, launch_free_list(e->stream, rider.stamp, rider.npages, rider.live, rider.freelist, rider.pool_state, rider.h_short));
        }
    }
    if (rc != SLAM_OK) return rc;
    // Resample gate: did the previous frame keep its population?  (Its verdict was made on the device; the host looks at it
    // only now, behind the launch above — a flag in mapped memory.)  Then the maps have not moved: the EKF runs IN PLACE on
    // the observed landmarks only, nothing is copied, the buffers do not flip.
    bool in_place = false;
    if (pf->gated && pf->has_anc) {
        int resampled = 1;
        if ((rc = slam_resample_happened_host(e, &resampled)) != SLAM_OK) return rc;
        in_place = !resampled;
        pf->frames_resampled += resampled ? 1 : 0;
    }
    if (comm) {
        // map rows of remote ancestors -> staging tail; issued behind the launch above, which does not need them
        if ((rc = finish_exchange(pf)) != SLAM_OK) {
            *collective_verdict = rc == SLAM_ERR_CAPACITY;
            return rc;
        }
    }
    // 3. per-landmark EKF (+ fused gather); the log-likelihood stays inside the engine for step 4
    const bool ekf = L > 0 && use_observations;
    const int mc = pf->map_cur, mn = 1 - mc;
    // Sharded: the ranks all-reduce the BLOCK maxima the weights' launch leaves in the engine (element by element: a few hundred
    // floats cost the wire what one costs) and the scan takes their maximum itself, as it does on one GPU — the maximum of the
    // same set of values, and one single-workgroup launch less per frame than reducing them to one float first
    // (SLAM_MAX_FINALIZE=1: that launch and a one-float all-reduce, as before).
    static const bool finalize = getenv("SLAM_MAX_FINALIZE") && atoi(getenv("SLAM_MAX_FINALIZE")) != 0;
    float* d_max = comm && finalize ? pf->d_max : nullptr;
    if (pf->split && !pf->paged && L > 0) {
        const int sc = pf->sp_cur;
        make_sio();   // (again: a layout move in front of the frame leaves other buffers than the ones the first look saw)
        if (ekf) {
            if (e->obs_nlandmarks != L) return SLAM_ERR_NOT_READY;
            if (sample_obs) SLAM_HIP_TRY(e, launch_obs_count(e->stream, e->d_obs_zx, e->d_obs_zy, L, d_hobs, ++pf->obs_seq_issued, pf->votes));
            // the particles' update (a frame that kept its population runs it out of place all the same: its gather index is
            // the identity) ...
            if (fused && comm) sio.group_filter = 2;   // the groups that waited for the exchange
            // (no rows received this frame: no group has an ancestor in the staging tail, the launch would find nothing to do)
            if (!fused || (comm && pf->rows_received > 0))
                if ((rc = slam_ekf_split_dev(e, pf->mean[sc], pf->mean[1 - sc], 2 * (int64_t)pf->Lp, pf->Lp, L, dst, dst + sn, dst + 2 * sn,
                                             anc, n, pf->cfg.meas_var, &sio)) != SLAM_OK)
                    return rc;
            pf->cstamp_now++;
            pf->sp_cur = 1 - sc;
            // ... then the classes' update, in place, once per class still in use
            rc = weights_with_classes(pf, L, true, d_max);
        } else {
            if (anc) {   // means and classes follow their particles
                const ProfScope prof(e, SLAM_PROF_PAGES);
                SLAM_HIP_TRY(e, launch_split_gather(e->stream, pf->mean[sc], pf->mean[1 - sc], pf->cls[sc], pf->cls[1 - sc], pf->Lp, anc, n,
                                                    pf->cstamp, ++pf->cstamp_now));
                pf->sp_cur = 1 - sc;
                rc = weights_with_classes(pf, 0, false, d_max);   // no observations: the list of classes in use only
            } else
                rc = slam_logweight_dev(e, pf->score, nullptr, pf->cfg.score_gain, n, pf->logw, d_max);
        }
    } else if (pf->paged) {
        const int pc = pf->pt_cur;
        if (ekf) {
            // touched pages of this frame's observation table, the update into fresh pages, the next frame's free list
            if (e->obs_nlandmarks != L) return SLAM_ERR_NOT_READY;
            SLAM_HIP_TRY(e, e->ll_buf.ensure(sizeof(float) * sn));
            if (!paged_listed) {
                if (int rc1 = issue_page_list()) return rc1;
                // a new free list when the old one runs short (decided on the device; the pages in use carry the last stamp)
                const ProfScope prof(e, SLAM_PROF_PAGES);
                SLAM_HIP_TRY(e, launch_free_list(e->stream, pf->stamp, pf->npages, pf->stamp_now, pf->freelist, pstate,
                                                 reinterpret_cast<int32_t*>(pf->d_hres) + 20));
            }
            PagedEkfArgs a;
            a.ol.id = lo.id;
            a.ol.zx = lo.zx;
            a.ol.zy = lo.zy;
            a.ol.round = lo.round;
            a.ol.count = lo.count;
            a.tmask = tmask;
            a.tbase = tbase;
            a.pool = pf->pool;
            if (pf->split) {   // split pages: mean pages of two planes, the covariances per class
                a.geom = split_geom(pf);
                a.pool = split_pool(pf);
                a.cov = pf->cov;
                a.covx = pf->covx;
                a.plane_stride = pf->Lp;
                a.cls_in = pf->cls[pf->sp_cur];
                a.cls_out = pf->cls[1 - pf->sp_cur];
                a.cstamp = pf->cstamp;
                a.cstamp_now = pf->cstamp_now + 1;
            }
            a.pt_in = pf->pt[pc];
            a.pt_out = pf->pt[1 - pc];
            a.nb = pf->nb;
            a.anc = anc;
            a.n = n;
            a.nlandmarks = L;
            a.x = dst;
            a.y = dst + sn;
            a.th = dst + 2 * sn;
            a.obs_zx = e->d_obs_zx;
            a.obs_zy = e->d_obs_zy;
            a.meas_var = pf->cfg.meas_var;
            a.loglik = e->ll_buf.as<float>();
            a.loglik_user = nullptr;
            a.tpage = tpage;
            a.tindex = tindex;
            a.count = count;
            a.freelist = pf->freelist;
            a.pool_state = pstate;
            a.stamp = pf->stamp;
            a.stamp_now = ++pf->stamp_now;
            // pages staged per pass = what the last frames touched (a hint in mapped memory, read without waiting)
            SLAM_HIP_TRY(e, launch_ekf_paged(e->stream, a, e->prof_next(SLAM_PROF_EKF), form,
                                             __atomic_load_n(reinterpret_cast<int32_t*>(pf->h_res) + 30, __ATOMIC_RELAXED)));
            e->ll_n = n;
            pf->pt_cur = 1 - pc;
            if (pf->split) {   // the classes went with their particles; their covariances, once per class, with the weights
                pf->cstamp_now++;
                pf->sp_cur = 1 - pf->sp_cur;
                rc = weights_with_classes(pf, L, true, d_max);
            } else
                rc = slam_logweight_ekf_dev(e, pf->score, pf->cfg.score_gain, n, pf->logw, d_max);
        } else {
            if (anc) {   // the tables follow their particles
                const ProfScope prof(e, SLAM_PROF_PAGES);
                SLAM_HIP_TRY(e, launch_page_table_gather(e->stream, pf->pt[pc], pf->pt[1 - pc], pf->nb, anc, n, pf->stamp,
                                                         ++pf->stamp_now));
                pf->pt_cur = 1 - pc;
                if (pf->split) {   // ... and so do the classes
                    SLAM_HIP_TRY(e, launch_class_gather(e->stream, pf->cls[pf->sp_cur], pf->cls[1 - pf->sp_cur], anc, n, pf->cstamp,
                                                        ++pf->cstamp_now));
                    pf->sp_cur = 1 - pf->sp_cur;
                }
            }
            rc = anc && pf->split ? weights_with_classes(pf, 0, false, d_max)
                                  : slam_logweight_dev(e, pf->score, nullptr, pf->cfg.score_gain, n, pf->logw, d_max);
        }
    } else if (ekf && in_place) {
        if (sample_obs) SLAM_HIP_TRY(e, launch_obs_count(e->stream, e->d_obs_zx, e->d_obs_zy, L, d_hobs, ++pf->obs_seq_issued, pf->votes));
        rc = slam_ekf_update_dev(e, pf->map[mc], pf->map[mc], 5 * (int64_t)pf->Lp, pf->Lp, L, dst, dst + sn, dst + 2 * sn, nullptr, n,
                                 pf->cfg.meas_var, nullptr);
        if (rc != SLAM_OK) return rc;
        rc = slam_logweight_ekf_dev(e, pf->score, pf->cfg.score_gain, n, pf->logw, d_max);
    } else if (ekf) {
        if (sample_obs) SLAM_HIP_TRY(e, launch_obs_count(e->stream, e->d_obs_zx, e->d_obs_zy, L, d_hobs, ++pf->obs_seq_issued, pf->votes));
        rc = fused ? SLAM_OK   // the update went out with the score
                   : slam_ekf_update_dev(e, pf->map[mc], pf->map[mn], 5 * (int64_t)pf->Lp, pf->Lp, L, dst, dst + sn, dst + 2 * sn, anc,
                                         n, pf->cfg.meas_var, nullptr);
        if (rc != SLAM_OK) return rc;
        pf->map_cur = mn;
        rc = slam_logweight_ekf_dev(e, pf->score, pf->cfg.score_gain, n, pf->logw, d_max);
    } else {
        if (L > 0 && anc && !in_place) {   // the maps follow their particles even without an observation
            rc = slam_gather_map_dev(e, pf->map[mc], pf->map[mn], 5 * (int64_t)pf->Lp, 5 * (int64_t)pf->Lp, pf->Lp, pf->Lp, L,
                                     anc, n);
            if (rc != SLAM_OK) return rc;
            pf->map_cur = mn;
        }
        rc = slam_logweight_dev(e, pf->score, nullptr, pf->cfg.score_gain, n, pf->logw, d_max);
    }
    if (rc != SLAM_OK) return rc;
    // 4. weights: the maximum over all ranks, then fixed-point weights scanned as they are produced
    if (comm) {
        if (d_max) {
            if ((rc = comm_all_reduce_max_f32(comm, pf->d_max, 1)) != SLAM_OK) return rc;
        } else {
            if (e->bmax_n != n || e->bmax_count <= 0) return SLAM_ERR_NOT_READY;
            if ((rc = comm_all_reduce_max_f32(comm, e->bmax_buf.as<float>(), e->bmax_count)) != SLAM_OK) return rc;
        }
    }
    if ((rc = slam_quantise_scan_dev(e, pf->logw, d_max, n, comm ? pf->d_sum : nullptr)) != SLAM_OK) return rc;
    // 5. resample on the integer CDF
    if (!comm) {
        if ((rc = slam_ancestors_from_scan_dev(e, n, pf->cfg.seed, pf->frame, pf->anc[nxt])) != SLAM_OK) return rc;
    } else {
        // shard totals (with the gate: total, sum v, sum v^2 per rank)
        if ((rc = comm_all_gather(comm, pf->d_sum, pf->totals, (pf->gated ? 3 : 1) * sizeof(uint64_t))) != SLAM_OK) return rc;
        if ((rc = slam_offspring_from_scan_sharded_dev(e, n, pf->totals, pf->rank, pf->world, pf->cfg.seed, pf->frame,
                                                       pf->n_total, pf->first)) != SLAM_OK)
            return rc;
        // the "all-gather of surviving indices" (4 B x N_total) and, grouped into the same RCCL launch, this frame's poses
        // to every rank (12 B x N_total) for the next frame's motion + score — that launch then needs nothing from the exchange
        if ((rc = comm_all_gather2(comm, pf->first, pf->first_all, sn * sizeof(int32_t), dst, pf->pose_all,
                                   3 * sn * sizeof(float))) != SLAM_OK)
            return rc;
        // 6. gather index of every slot (remote ancestors -> rows of the staging tail) and the exchange plan, on the
        // device; the exchange itself happens at the start of the next frame, behind its motion + score launch
        if ((rc = slam_ancestors_sharded_dev(e, pf->first_all, pf->n_total, n, pf->rank, pf->world, pf->anc[nxt], pf->d_plan,
                                             pf->pose_idx[nxt])) != SLAM_OK)
            return rc;
        pf->exchange_pending = true;
    }
    pf->cur = nxt;
    pf->has_anc = true;
    pf->last_ekf = ekf;
    pf->frame++;
    return SLAM_OK;
}

int slam_pf_rows_received(const slam_pf* pf) { return pf ? pf->rows_received : 0; }

int64_t slam_pf_frames_resampled(const slam_pf* pf) { return pf ? pf->frames_resampled : 0; }

int64_t slam_pf_layout_changes(const slam_pf* pf) { return pf ? pf->conversions : 0; }

int slam_pf_device_view(slam_pf* pf, slam_pf_view* out)
{
    if (!pf || !out) return SLAM_ERR_INVALID_ARG;
    out->pose = pf->pose[pf->cur];
    const bool rows = pf->L && !pf->paged && !pf->split;   // pages and split maps have no rows to look at: slam_pf_set_map_dev
    out->map = rows ? pf->map[pf->map_cur] : nullptr;
    out->map_spare = rows ? pf->map[1 - pf->map_cur] : nullptr;
    out->anc = pf->has_anc ? pf->anc[pf->cur] : nullptr;
    out->row_stride = 5 * (int64_t)pf->Lp;
    out->plane_stride = pf->Lp;
    out->map_rows = pf->cap;
    out->score = pf->has_anc ? pf->score : nullptr;
    out->logw = pf->has_anc ? pf->logw : nullptr;
    out->loglik = pf->has_anc && pf->last_ekf && pf->e->ll_n == pf->n ? pf->e->ll_buf.as<float>() : nullptr;
    out->count = pf->has_anc ? pf->count : nullptr;
    return SLAM_OK;
}

int slam_pf_layout(const slam_pf* pf)
{
    if (!pf || !pf->L) return SLAM_MAP_ROWS;
    return pf->paged ? (pf->split ? SLAM_MAP_SPLIT_PAGES : SLAM_MAP_PAGES) : pf->split ? SLAM_MAP_SPLIT : SLAM_MAP_ROWS;
}

int slam_pf_split_device_view(slam_pf* pf, slam_pf_split_view* out)
{
    if (!pf || !out) return SLAM_ERR_INVALID_ARG;
    if (!pf->split || !pf->L) return SLAM_ERR_NOT_READY;
    out->mean = pf->paged ? nullptr : pf->mean[pf->sp_cur];   // split pages: the means are on pages (slam_pf_paged_device_view)
    out->cov = pf->cov;
    out->cls = pf->cls[pf->sp_cur];
    out->live = pf->live[pf->live_cur];
    out->live_count = pf->cov_cnt + pf->cov_phase;
    out->plane_stride = pf->Lp;
    out->rows = pf->cap;
    return SLAM_OK;
}

int slam_pf_paged_device_view(slam_pf* pf, slam_pf_paged_view* out)
{
    if (!pf || !out) return SLAM_ERR_INVALID_ARG;
    if (!pf->paged) return SLAM_ERR_NOT_READY;
    const PageGeom g = pf->split ? split_geom(pf) : PageGeom();
    out->pool = pf->split ? split_pool(pf) : pf->pool;
    out->planes = g.planes;
    out->reserved = 0;
    out->half_pages = pf->split ? g.half_pages : (int64_t)pf->npages;
    out->gap_floats = g.gap;
    out->table = pf->pt[pf->pt_cur];
    out->freelist = pf->freelist;
    out->state = pf->page_scratch;
    out->stamp = pf->stamp;
    out->stamp_now = pf->stamp_now;
    out->page_landmarks = kPageLandmarks;
    out->pages_per_particle = pf->nb;
    out->table_rows = pf->cap;
    out->npages = pf->npages;
    return SLAM_OK;
}

static int slam_pf_best_impl(slam_pf* pf, float pose[3], float* logw, int32_t* index);

int slam_pf_best(slam_pf* pf, float pose[3], float* logw, int32_t* index)
{
    return collective_result(pf, slam_pf_best_impl(pf, pose, logw, index));
}

static int slam_pf_best_impl(slam_pf* pf, float pose[3], float* logw, int32_t* index)
{
    if (!pf || !pose) return SLAM_ERR_INVALID_ARG;
    slam_engine* e = pf->e;
    SLAM_HIP_TRY(e, hipSetDevice(e->device));
    // the log-weights of the last frame belong to pose[cur] BEFORE the pending gather
    const float* p = pf->pose[pf->cur];
    const size_t sn = (size_t)pf->n;
    float r[5];
    if (!pf->comm) {   // one launch, the result lands in mapped host memory: no copy, no stream synchronisation
        const uint32_t seq = ++pf->res_seq;
        SLAM_HIP_TRY(e, launch_best_particle(e->stream, pf->logw, pf->n, p, p + sn, p + 2 * sn, 0, pf->res_dev, pf->d_hres,
                                             reinterpret_cast<uint32_t*>(pf->d_hres + 16), seq));
        if (int rc = wait_result(pf, seq)) return rc;
        memcpy(r, pf->h_res, sizeof r);
    } else {   // every rank's candidate to every rank; the first maximum = the lowest rank = the lowest id
        SLAM_HIP_TRY(e, launch_best_particle(e->stream, pf->logw, pf->n, p, p + sn, p + 2 * sn, (int64_t)pf->rank * pf->n,
                                             pf->res_dev, nullptr, nullptr, 0));
        if (int rc = comm_all_gather(pf->comm, pf->res_dev, pf->res_all, 5 * sizeof(float))) return rc;
        std::vector<float> all(5 * (size_t)pf->world);
        SLAM_HIP_TRY(e, hipMemcpyAsync(all.data(), pf->res_all, all.size() * 4, hipMemcpyDeviceToHost, e->stream));
        if (int rc = comm_wait_stream(pf->comm)) return rc;
        int best = 0;
        for (int q = 1; q < pf->world; ++q)
            if (all[5 * q] > all[5 * best]) best = q;
        memcpy(r, &all[5 * best], sizeof r);
    }
    int32_t gid;
    memcpy(&gid, &r[1], 4);
    for (int k = 0; k < 3; ++k) pose[k] = r[2 + k];
    if (logw) *logw = r[0];
    if (index) *index = gid;
    return SLAM_OK;
}

static int slam_pf_mean_impl(slam_pf* pf, float ref_theta, float pose[3]);

int slam_pf_mean(slam_pf* pf, float ref_theta, float pose[3])
{
    return collective_result(pf, slam_pf_mean_impl(pf, ref_theta, pose));
}

static int slam_pf_mean_impl(slam_pf* pf, float ref_theta, float pose[3])
{
    if (!pf || !pose) return SLAM_ERR_INVALID_ARG;
    slam_engine* e = pf->e;
    SLAM_HIP_TRY(e, hipSetDevice(e->device));
    const size_t sn = (size_t)pf->n;
    const float *x = pf->pose[pf->cur], *y = x + sn, *th = x + 2 * sn;
    const int32_t* idx = pf->has_anc ? pf->anc[pf->cur] : nullptr;
    if (pf->comm && pf->has_anc) {   // the ancestors' poses are in the all-gathered array
        if (int rc = comm_all_gather_finish(pf->comm)) return rc;
        x = pf->pose_all;
        y = x + sn;
        th = x + 2 * sn;
        idx = pf->pose_idx[pf->cur];
    }
    unsigned int* ticket = reinterpret_cast<unsigned int*>(pf->sums_acc + 4);
    long long sums[4] = { 0, 0, 0, 0 };
    if (!pf->comm) {
        const uint32_t seq = ++pf->res_seq;
        SLAM_HIP_TRY(e, launch_pose_sums(e->stream, x, y, th, idx, pf->n, ref_theta, pf->sums_acc, ticket,
                                         reinterpret_cast<long long*>(pf->res_dev), reinterpret_cast<long long*>(pf->d_hres),
                                         reinterpret_cast<uint32_t*>(pf->d_hres + 16), seq));
        if (int rc = wait_result(pf, seq)) return rc;
        memcpy(sums, pf->h_res, sizeof sums);
    } else {   // integer sums: adding the ranks' shares in any order gives the single-GPU bits
        SLAM_HIP_TRY(e, launch_pose_sums(e->stream, x, y, th, idx, pf->n, ref_theta, pf->sums_acc, ticket,
                                         reinterpret_cast<long long*>(pf->res_dev), nullptr, nullptr, 0));
        if (int rc = comm_all_gather(pf->comm, pf->res_dev, pf->res_all, 4 * sizeof(long long))) return rc;
        std::vector<long long> all(4 * (size_t)pf->world);
        SLAM_HIP_TRY(e, hipMemcpyAsync(all.data(), pf->res_all, all.size() * 8, hipMemcpyDeviceToHost, e->stream));
        if (int rc = comm_wait_stream(pf->comm)) return rc;
        for (int q = 0; q < pf->world; ++q)
            for (int k = 0; k < 4; ++k) sums[k] += all[4 * (size_t)q + k];
    }
    const double nt = (double)pf->n_total;
    pose[0] = (float)((double)sums[0] / 4294967296.0 / nt);
    pose[1] = (float)((double)sums[1] / 4294967296.0 / nt);
    pose[2] = (float)((double)ref_theta + atan2((double)sums[2], (double)sums[3]));
    return SLAM_OK;
}

static int slam_pf_get_poses_host_impl(slam_pf* pf, float* x, float* y, float* theta);

int slam_pf_get_poses_host(slam_pf* pf, float* x, float* y, float* theta)
{
    return collective_result(pf, slam_pf_get_poses_host_impl(pf, x, y, theta));
}

static int slam_pf_get_poses_host_impl(slam_pf* pf, float* x, float* y, float* theta)
{
    if (!pf || !x || !y || !theta) return SLAM_ERR_INVALID_ARG;
    const size_t n = (size_t)pf->n;
    float* tmp = pf->pose[1 - pf->cur];   // the other buffer is free between frames
    float* out[3] = { x, y, theta };
    if (pf->comm && pf->has_anc) {   // the ancestors' poses are in the all-gathered array
        if (int rc = comm_all_gather_finish(pf->comm)) return rc;
        for (int k = 0; k < 3; ++k)
            if (int rc = gathered_copy_out(pf, pf->pose_all + k * n, pf->pose_idx[pf->cur], out[k], tmp)) return rc;
        return SLAM_OK;
    }
    const float* p = pf->pose[pf->cur];
    for (int k = 0; k < 3; ++k)
        if (int rc = gathered_copy_out(pf, p + k * n, pf->has_anc ? pf->anc[pf->cur] : nullptr, out[k], tmp)) return rc;
    return SLAM_OK;
}

static int slam_pf_get_map_host_impl(slam_pf* pf, float* rows);

int slam_pf_get_map_host(slam_pf* pf, float* rows)
{
    return collective_result(pf, slam_pf_get_map_host_impl(pf, rows));
}

static int slam_pf_get_map_host_impl(slam_pf* pf, float* rows)
{
    if (!pf || !rows || !pf->L) return SLAM_ERR_INVALID_ARG;
    const size_t n = (size_t)pf->n, L = (size_t)pf->L, Lp = (size_t)pf->Lp;
    if (pf->comm)
        if (int rc = finish_exchange(pf)) return rc;   // collective: remote ancestors' rows into the staging tail
    if (pf->paged || pf->split) {   // -> rows in a scratch buffer (the pending gather applied on the way), then the copy
        float* dense = nullptr;
        if (hipMalloc((void**)&dense, 5 * Lp * n * 4) != hipSuccess) return SLAM_ERR_HIP;
        int rc = SLAM_OK;
        const int32_t* idx = pf->has_anc ? pf->anc[pf->cur] : nullptr;
        if ((split_pages(pf) ? launch_rows_from_split_pages(pf->e->stream, split_pool(pf), split_geom(pf), pf->pt[pf->pt_cur], pf->nb, pf->cov,
                                                            pf->cls[pf->sp_cur], pf->Lp, idx, pf->n, dense, 5 * (int64_t)Lp, pf->Lp, pf->L)
             : pf->split ? launch_rows_from_split(pf->e->stream, pf->mean[pf->sp_cur], pf->cov, pf->cls[pf->sp_cur], pf->Lp, idx, pf->n, dense,
                                                5 * (int64_t)Lp, pf->Lp, pf->L)
                       : launch_rows_from_pages(pf->e->stream, pf->pool, pf->pt[pf->pt_cur], pf->nb, idx, pf->n, dense, 5 * (int64_t)Lp,
                                                pf->Lp, pf->L)) != hipSuccess)
            rc = SLAM_ERR_HIP;
        if (rc == SLAM_OK) rc = slam_engine_sync(pf->e);
        if (rc == SLAM_OK && hipMemcpy2D(rows, L * 4, dense, Lp * 4, L * 4, 5 * n, hipMemcpyDeviceToHost) != hipSuccess)
            rc = SLAM_ERR_HIP;
        (void)hipFree(dense);
        return rc;
    }
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

static int slam_pf_get_map_rows_host_impl(slam_pf* pf, const int32_t* particle, int count, float* rows);

int slam_pf_get_map_rows_host(slam_pf* pf, const int32_t* particle, int count, float* rows)
{
    return collective_result(pf, slam_pf_get_map_rows_host_impl(pf, particle, count, rows));
}

static int slam_pf_get_map_rows_host_impl(slam_pf* pf, const int32_t* particle, int count, float* rows)
{
    if (!pf || !pf->L || count < 0 || (count > 0 && (!particle || !rows))) return SLAM_ERR_INVALID_ARG;
    for (int k = 0; k < count; ++k)
        if (particle[k] < 0 || particle[k] >= pf->n) return SLAM_ERR_INVALID_ARG;
    slam_engine* e = pf->e;
    if (pf->comm)
        if (int rc = finish_exchange(pf)) return rc;   // collective: remote ancestors' rows into the staging tail
    if (count == 0) return SLAM_OK;
    SLAM_HIP_TRY(e, hipSetDevice(e->device));
    if (pf->sel_cap < count) {
        SLAM_HIP_TRY(e, hipStreamSynchronize(e->stream));
        if (pf->sel) (void)hipFree(pf->sel);
        pf->sel = nullptr;
        pf->sel_cap = 0;
        SLAM_HIP_TRY(e, hipMalloc((void**)&pf->sel, 2 * sizeof(int32_t) * (size_t)count));
        pf->sel_cap = count;
    }
    const size_t L = (size_t)pf->L, Lp = (size_t)pf->Lp;
    float* dense = nullptr;
    SLAM_HIP_TRY(e, hipMalloc((void**)&dense, 5 * Lp * (size_t)count * 4));
    int32_t *sel = pf->sel, *src = pf->sel + pf->sel_cap;
    int rc = SLAM_OK;
    auto ok = [&](hipError_t err, const char* what) {
        if (err != hipSuccess && rc == SLAM_OK) rc = slam_engine_fail_hip(e, err, what);
        return err == hipSuccess;
    };
    if (ok(hipMemcpyAsync(sel, particle, sizeof(int32_t) * (size_t)count, hipMemcpyHostToDevice, e->stream), "copy of the particle list") &&
        ok(launch_compose_index(e->stream, sel, pf->has_anc ? pf->anc[pf->cur] : nullptr, count, src), "compose_index")) {
        if (split_pages(pf))
            ok(launch_rows_from_split_pages(e->stream, split_pool(pf), split_geom(pf), pf->pt[pf->pt_cur], pf->nb, pf->cov, pf->cls[pf->sp_cur],
                                            pf->Lp, src, count, dense, 5 * (int64_t)Lp, pf->Lp, pf->L), "rows_from_split_pages");
        else if (pf->split)
            ok(launch_rows_from_split(e->stream, pf->mean[pf->sp_cur], pf->cov, pf->cls[pf->sp_cur], pf->Lp, src, count, dense,
                                      5 * (int64_t)Lp, pf->Lp, pf->L), "rows_from_split");
        else if (pf->paged)
            ok(launch_rows_from_pages(e->stream, pf->pool, pf->pt[pf->pt_cur], pf->nb, src, count, dense, 5 * (int64_t)Lp, pf->Lp, pf->L),
               "rows_from_pages");
        else
            ok(launch_gather_map(e->stream, pf->map[pf->map_cur], dense, 5 * (int64_t)Lp, 5 * (int64_t)Lp, pf->Lp, pf->Lp, pf->L, src,
                                 count), "gather_map");
    }
    if (rc == SLAM_OK) ok(hipStreamSynchronize(e->stream), "hipStreamSynchronize");
    if (rc == SLAM_OK) ok(hipMemcpy2D(rows, L * 4, dense, Lp * 4, L * 4, 5 * (size_t)count, hipMemcpyDeviceToHost), "hipMemcpy2D");
    (void)hipFree(dense);
    return rc;
}

}  // extern "C"
