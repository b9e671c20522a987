// comm.hip — communicators of the sharded particle filter (see comm.h): RCCL over xGMI, and the in-process
// transport.  The C ABI half (creation / destruction) is declared in include/slam_hip.h.

#include "comm.h"

#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>

#include "engine_internal.h"

using namespace slam;

// ------------------------------------------------------------------ in-process group (rehearsal / single-process hosts)
struct slam_local_group {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;
    double timeout_s = 120.0;   // how long a rank waits for the others (SLAM_COMM_TIMEOUT_S)
    // what every rank published for the collective in flight
    const void* send[kMaxRanks] = {};
    int64_t cnt[kMaxRanks][kMaxRanks] = {};   // all-to-all: cnt[src][dst] floats
    float fmax[kMaxRanks][4] = {};
    std::vector<float> fbig[kMaxRanks];   // all-reduce MAX of longer arrays (the ranks' block maxima of the log-weights)

    // reusable barrier; false when a rank failed to arrive within the time limit (the group is then broken for good)
    bool barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return false;
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
            return true;
        }
        if (!cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return generation != gen || broken; })) {
            broken = true;
            cv.notify_all();
            return false;
        }
        return !broken;
    }
    // a rank that cannot go on says so: everybody waiting (now or later) fails at once instead of timing out
    void break_group()
    {
        std::lock_guard<std::mutex> lk(m);
        broken = true;
        cv.notify_all();
    }
    bool is_broken()
    {
        std::lock_guard<std::mutex> lk(m);
        return broken;
    }
};

struct slam_comm {
    slam_engine* e = nullptr;
    int rank = 0, world = 1;
    ncclComm_t nccl = nullptr;            // RCCL transport
    slam_local_group* group = nullptr;    // in-process transport
    // Every collective is issued on the ENGINE's stream, in program order: no second stream, no event hand-overs
    // (measured on a one-rank communicator at 64k x 500: a dedicated communicator stream with fork / join events
    // around each of the four small collectives of a frame made the frame 0.34 ms instead of 0.22 ms single-GPU).  The price: the
    // all-gather of the poses does not run beside the EKF.
    bool async_pending = false;
    // Failure path.  `dead`: this communicator was aborted — by this rank after an error of its own (slam_comm_abort, a
    // failed slam_pf_step) or because a poll found an asynchronous RCCL error / a broken group / a wait past the time
    // limit.  Every later call fails at once with SLAM_ERR_COMM.  Aborting is what releases the PEERS: ncclCommAbort makes
    // the RCCL kernels of this rank leave, the peers' kernels then see their partner gone and their own polls
    // (ncclCommGetAsyncError, the time limit) turn that into SLAM_ERR_COMM instead of a hang.
    bool dead = false;
    double timeout_s = 120.0;   // SLAM_COMM_TIMEOUT_S: longest a host-side wait of a sharded session may last
};

namespace {

int fail_nccl(slam_comm* c, ncclResult_t r, const char* what)
{
    snprintf(c->e->err, sizeof c->e->err, "%s: %s", what, ncclGetErrorString(r));
    return SLAM_ERR_COMM;
}

#define NCCL_TRY(c, call)                                          \
    do {                                                           \
        ncclResult_t r__ = (call);                                 \
        if (r__ != ncclSuccess) return fail_nccl((c), r__, #call); \
    } while (0)

#define CHIP_TRY(c, call) SLAM_HIP_TRY((c)->e, call)
// inside the in-process transport: a HIP failure of one rank must not leave the others waiting at the next rendezvous
#define LHIP_TRY(c, call)                                                  \
    do {                                                                   \
        hipError_t err__ = (call);                                         \
        if (err__ != hipSuccess) {                                         \
            (c)->group->break_group();                                     \
            (c)->dead = true;                                              \
            return slam_engine_fail_hip((c)->e, err__, #call);             \
        }                                                                  \
    } while (0)

int local_fail(slam_comm* c, const char* what)
{
    c->group->break_group();   // the peers must not wait for a rank that has given up
    c->dead = true;
    snprintf(c->e->err, sizeof c->e->err, "in-process group: %s (a rank did not arrive or gave up)", what);
    return SLAM_ERR_COMM;
}

int dead_fail(slam_comm* c)
{
    snprintf(c->e->err, sizeof c->e->err, "communicator aborted earlier (after an error on this or another rank)");
    return SLAM_ERR_COMM;
}

double env_timeout()
{
    const char* v = getenv("SLAM_COMM_TIMEOUT_S");
    const double t = v ? atof(v) : 0.0;
    return t > 0.0 ? t : 120.0;
}

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// in-process all-gather: publish, rendezvous, pull every rank's block, rendezvous again (nobody may reuse its send
// buffer before every peer has copied it)
int local_all_gather(slam_comm* c, const void* d_send, void* d_recv, size_t bytes)
{
    slam_local_group* g = c->group;
    LHIP_TRY(c, hipStreamSynchronize(c->e->stream));
    g->send[c->rank] = d_send;
    if (!g->barrier()) return local_fail(c, "all_gather");
    for (int q = 0; q < c->world; ++q)
        LHIP_TRY(c, hipMemcpyAsync(static_cast<char*>(d_recv) + (size_t)q * bytes, g->send[q], bytes, hipMemcpyDefault,
                                   c->e->stream));
    LHIP_TRY(c, hipStreamSynchronize(c->e->stream));
    if (!g->barrier()) return local_fail(c, "all_gather");
    return SLAM_OK;
}

}  // namespace

namespace slam {

int comm_rank(const slam_comm* c) { return c->rank; }
int comm_world(const slam_comm* c) { return c->world; }
slam_engine* comm_engine(const slam_comm* c) { return c->e; }

// ---- failure path
int comm_abort(slam_comm* c)
{
    if (!c || c->dead) return SLAM_OK;
    c->dead = true;
    if (c->group) c->group->break_group();
    if (c->nccl) {
        (void)ncclCommAbort(c->nccl);   // frees the communicator; its kernels on the device leave
        c->nccl = nullptr;
    }
    return SLAM_OK;
}

int comm_poll(slam_comm* c)
{
    if (c->dead) return dead_fail(c);
    if (c->group) {
        if (c->group->is_broken()) return local_fail(c, "a rank failed");
        return SLAM_OK;
    }
    ncclResult_t st = ncclSuccess;
    const ncclResult_t r = ncclCommGetAsyncError(c->nccl, &st);
    if (r == ncclSuccess && (st == ncclSuccess || st == ncclInProgress)) return SLAM_OK;
    const int rc = fail_nccl(c, r != ncclSuccess ? r : st, "asynchronous RCCL error");
    comm_abort(c);
    return rc;
}

// hipStreamSynchronize for a stream that may hold a collective: polls the stream and the communicator, bounded in time
int comm_wait_stream(slam_comm* c)
{
    const double t0 = now_s();
    for (long spin = 0;; ++spin) {
        const hipError_t q = hipStreamQuery(c->e->stream);
        if (q == hipSuccess) return SLAM_OK;
        if (q != hipErrorNotReady) {   // this rank cannot go on: release the peers before reporting
            comm_abort(c);
            return slam_engine_fail_hip(c->e, q, "hipStreamQuery");
        }
        if ((spin & 1023) == 1023) {
            if (int rc = comm_poll(c)) return rc;
            if (now_s() - t0 > c->timeout_s) {
                comm_abort(c);
                snprintf(c->e->err, sizeof c->e->err, "a collective did not finish within %.0f s (SLAM_COMM_TIMEOUT_S): aborted", c->timeout_s);
                return SLAM_ERR_COMM;
            }
        }
    }
}

// wait for `*flag == seq` (mapped host memory written by a kernel that may sit behind a collective)
int comm_wait_flag(slam_comm* c, const volatile uint32_t* flag, uint32_t seq)
{
    const double t0 = now_s();
    for (long spin = 0;; ++spin) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return SLAM_OK;
        if ((spin & 0xfffff) == 0xfffff) {
            if (int rc = comm_poll(c)) return rc;
            if (now_s() - t0 > c->timeout_s) {
                comm_abort(c);
                snprintf(c->e->err, sizeof c->e->err, "a result did not arrive within %.0f s (SLAM_COMM_TIMEOUT_S): aborted", c->timeout_s);
                return SLAM_ERR_COMM;
            }
        }
    }
}

int comm_all_reduce_max_f32(slam_comm* c, float* d_buf, int count)
{
    if (c->dead) return dead_fail(c);
    const ProfScope prof(c->e, SLAM_PROF_COLLECTIVES);
    if (count <= 0) return SLAM_OK;
    if (c->group && count > 4) {   // (the same exchange through per-rank vectors)
        slam_local_group* g = c->group;
        std::vector<float>& mine = g->fbig[c->rank];
        mine.resize((size_t)count);
        LHIP_TRY(c, hipMemcpyAsync(mine.data(), d_buf, sizeof(float) * count, hipMemcpyDeviceToHost, c->e->stream));
        LHIP_TRY(c, hipStreamSynchronize(c->e->stream));
        if (!g->barrier()) return local_fail(c, "all_reduce");
        std::vector<float> out(mine);
        for (int q = 0; q < c->world; ++q) {
            if ((int)g->fbig[q].size() != count) return local_fail(c, "all_reduce (counts differ)");
            for (int k = 0; k < count; ++k)
                if (g->fbig[q][k] > out[k]) out[k] = g->fbig[q][k];
        }
        if (!g->barrier()) return local_fail(c, "all_reduce");   // everybody has read everybody's vector
        LHIP_TRY(c, hipMemcpyAsync(d_buf, out.data(), sizeof(float) * count, hipMemcpyHostToDevice, c->e->stream));
        LHIP_TRY(c, hipStreamSynchronize(c->e->stream));   // `out` is a local buffer
        return SLAM_OK;
    }
    if (c->group) {
        slam_local_group* g = c->group;
        float mine[4];
        LHIP_TRY(c, hipMemcpyAsync(mine, d_buf, sizeof(float) * count, hipMemcpyDeviceToHost, c->e->stream));
        LHIP_TRY(c, hipStreamSynchronize(c->e->stream));
        for (int k = 0; k < count; ++k) g->fmax[c->rank][k] = mine[k];
        if (!g->barrier()) return local_fail(c, "all_reduce");
        for (int q = 0; q < c->world; ++q)
            for (int k = 0; k < count; ++k)
                if (g->fmax[q][k] > mine[k]) mine[k] = g->fmax[q][k];
        LHIP_TRY(c, hipMemcpyAsync(d_buf, mine, sizeof(float) * count, hipMemcpyHostToDevice, c->e->stream));
        LHIP_TRY(c, hipStreamSynchronize(c->e->stream));   // `mine` is a stack buffer
        if (!g->barrier()) return local_fail(c, "all_reduce");
        return SLAM_OK;
    }
    NCCL_TRY(c, ncclAllReduce(d_buf, d_buf, (size_t)count, ncclFloat, ncclMax, c->nccl, c->e->stream));
    return SLAM_OK;
}

int comm_all_gather(slam_comm* c, const void* d_send, void* d_recv, size_t bytes)
{
    if (c->dead) return dead_fail(c);
    const ProfScope prof(c->e, SLAM_PROF_COLLECTIVES);
    if (bytes == 0) return SLAM_OK;
    if (c->group) return local_all_gather(c, d_send, d_recv, bytes);
    NCCL_TRY(c, ncclAllGather(d_send, d_recv, bytes, ncclChar, c->nccl, c->e->stream));
    return SLAM_OK;
}

// Two all-gathers as ONE grouped RCCL operation (aggregated into a single launch: one latency instead of two).
int comm_all_gather2(slam_comm* c, const void* d_send_a, void* d_recv_a, size_t bytes_a, const void* d_send_b,
                     void* d_recv_b, size_t bytes_b)
{
    if (c->dead) return dead_fail(c);
    const ProfScope prof(c->e, SLAM_PROF_COLLECTIVES);
    if (c->group) {
        if (int rc = bytes_a ? local_all_gather(c, d_send_a, d_recv_a, bytes_a) : SLAM_OK) return rc;
        return bytes_b ? local_all_gather(c, d_send_b, d_recv_b, bytes_b) : SLAM_OK;
    }
    NCCL_TRY(c, ncclGroupStart());
    ncclResult_t bad = ncclSuccess;
    if (bytes_a) bad = ncclAllGather(d_send_a, d_recv_a, bytes_a, ncclChar, c->nccl, c->e->stream);
    if (bytes_b && bad == ncclSuccess) bad = ncclAllGather(d_send_b, d_recv_b, bytes_b, ncclChar, c->nccl, c->e->stream);
    const ncclResult_t end = ncclGroupEnd();   // always close the group
    if (bad != ncclSuccess) return fail_nccl(c, bad, "ncclAllGather (grouped)");
    if (end != ncclSuccess) return fail_nccl(c, end, "ncclGroupEnd");
    return SLAM_OK;
}

int comm_all_gather_begin(slam_comm* c, const void* d_send, void* d_recv, size_t bytes)
{
    if (c->dead) return dead_fail(c);
    const ProfScope prof(c->e, SLAM_PROF_COLLECTIVES);
    if (c->async_pending) return SLAM_ERR_INVALID_ARG;   // one asynchronous gather at a time
    if (bytes == 0) return SLAM_OK;
    if (c->group) return local_all_gather(c, d_send, d_recv, bytes);
    NCCL_TRY(c, ncclAllGather(d_send, d_recv, bytes, ncclChar, c->nccl, c->e->stream));
    c->async_pending = true;   // stream-ordered: comm_all_gather_finish has nothing to wait for
    return SLAM_OK;
}

int comm_all_gather_finish(slam_comm* c)
{
    c->async_pending = false;
    return SLAM_OK;
}

int comm_all_to_all_f32(slam_comm* c, const float* d_send, const int64_t* send_floats, float* d_recv,
                        const int64_t* recv_floats)
{
    if (c->dead) return dead_fail(c);
    const ProfScope prof(c->e, SLAM_PROF_COLLECTIVES);
    if (c->group) {
        slam_local_group* g = c->group;
        LHIP_TRY(c, hipStreamSynchronize(c->e->stream));
        g->send[c->rank] = d_send;
        for (int q = 0; q < c->world; ++q) g->cnt[c->rank][q] = send_floats[q];
        if (!g->barrier()) return local_fail(c, "all_to_all");
        int64_t roff = 0;
        for (int q = 0; q < c->world; ++q) {
            if (g->cnt[q][c->rank] != recv_floats[q]) return local_fail(c, "all_to_all split sizes disagree");   // breaks the group: the peers fail at once
            int64_t soff = 0;   // where my block starts inside rank q's send buffer
            for (int d = 0; d < c->rank; ++d) soff += g->cnt[q][d];
            if (recv_floats[q] > 0)
                LHIP_TRY(c, hipMemcpyAsync(d_recv + roff, static_cast<const float*>(g->send[q]) + soff,
                                           sizeof(float) * (size_t)recv_floats[q], hipMemcpyDefault, c->e->stream));
            roff += recv_floats[q];
        }
        LHIP_TRY(c, hipStreamSynchronize(c->e->stream));
        if (!g->barrier()) return local_fail(c, "all_to_all");
        return SLAM_OK;
    }
    // one grouped exchange: direct peer-to-peer transfers over xGMI, every link busy, no ring
    NCCL_TRY(c, ncclGroupStart());
    int64_t soff = 0, roff = 0;
    ncclResult_t first_bad = ncclSuccess;
    for (int q = 0; q < c->world; ++q) {
        if (send_floats[q] > 0 && first_bad == ncclSuccess)
            first_bad = ncclSend(d_send + soff, (size_t)send_floats[q], ncclFloat, q, c->nccl, c->e->stream);
        if (recv_floats[q] > 0 && first_bad == ncclSuccess)
            first_bad = ncclRecv(d_recv + roff, (size_t)recv_floats[q], ncclFloat, q, c->nccl, c->e->stream);
        soff += send_floats[q];
        roff += recv_floats[q];
    }
    const ncclResult_t end = ncclGroupEnd();   // always close the group, even after a failed call inside it
    if (first_bad != ncclSuccess) return fail_nccl(c, first_bad, "ncclSend/ncclRecv");
    if (end != ncclSuccess) return fail_nccl(c, end, "ncclGroupEnd");
    return SLAM_OK;
}

}  // namespace slam

// ------------------------------------------------------------------ C ABI
namespace {

int comm_common_init(slam_comm* c)
{
    slam_engine* e = c->e;
    SLAM_HIP_TRY(e, hipSetDevice(e->device));
    return SLAM_OK;
}

}  // namespace

extern "C" {

int slam_comm_unique_id(uint8_t id[SLAM_COMM_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == SLAM_COMM_ID_BYTES, "ncclUniqueId size");
    if (!id) return SLAM_ERR_INVALID_ARG;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return SLAM_ERR_COMM;
    memcpy(id, &u, sizeof u);
    return SLAM_OK;
}

int slam_comm_create_rccl(slam_engine* e, int rank, int world, const uint8_t id[SLAM_COMM_ID_BYTES], slam_comm** out)
{
    if (!e || !out || !id || world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return SLAM_ERR_INVALID_ARG;
    *out = nullptr;
    slam_comm* c = new (std::nothrow) slam_comm();
    if (!c) return SLAM_ERR_HIP;
    c->e = e;
    c->rank = rank;
    c->world = world;
    c->timeout_s = env_timeout();
    if (int rc = comm_common_init(c)) {
        slam_comm_destroy(c);
        return rc;
    }
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    const ncclResult_t r = ncclCommInitRank(&c->nccl, world, u, rank);   // collective: every rank calls it
    if (r != ncclSuccess) {
        const int rc = fail_nccl(c, r, "ncclCommInitRank");
        c->nccl = nullptr;
        slam_comm_destroy(c);
        return rc;
    }
    e->comm = c;
    *out = c;
    return SLAM_OK;
}

int slam_local_group_create(int world, slam_local_group** out)
{
    if (!out || world < 1 || world > kMaxRanks) return SLAM_ERR_INVALID_ARG;
    slam_local_group* g = new (std::nothrow) slam_local_group();
    if (!g) return SLAM_ERR_HIP;
    g->world = world;
    g->timeout_s = env_timeout();
    *out = g;
    return SLAM_OK;
}

int slam_local_group_destroy(slam_local_group* g)
{
    delete g;
    return SLAM_OK;
}

int slam_comm_create_local(slam_engine* e, slam_local_group* g, int rank, slam_comm** out)
{
    if (!e || !g || !out || rank < 0 || rank >= g->world) return SLAM_ERR_INVALID_ARG;
    *out = nullptr;
    slam_comm* c = new (std::nothrow) slam_comm();
    if (!c) return SLAM_ERR_HIP;
    c->e = e;
    c->rank = rank;
    c->world = g->world;
    c->group = g;
    c->timeout_s = g->timeout_s;
    if (int rc = comm_common_init(c)) {
        slam_comm_destroy(c);
        return rc;
    }
    e->comm = c;
    *out = c;
    return SLAM_OK;
}

int slam_comm_abort(slam_comm* c)
{
    if (!c) return SLAM_ERR_INVALID_ARG;
    (void)hipSetDevice(c->e->device);
    return comm_abort(c);
}

int slam_comm_rank(const slam_comm* c) { return c ? c->rank : -1; }
int slam_comm_world(const slam_comm* c) { return c ? c->world : -1; }

int slam_comm_destroy(slam_comm* c)
{
    if (!c) return SLAM_OK;
    (void)hipSetDevice(c->e->device);
    if (!c->dead) (void)hipStreamSynchronize(c->e->stream);
    if (c->nccl) (void)ncclCommDestroy(c->nccl);
    if (c->e->comm == c) c->e->comm = nullptr;
    delete c;
    return SLAM_OK;
}

}  // extern "C"
