// comm.h — the exchange steps between the ranks of a sharded particle filter, as the engine issues them itself
// (no host framework between the launches).  Two transports behind one interface:
//   RCCL   one process (or host thread) per GPU, collectives over xGMI: the production transport;
//   local  all ranks live in ONE process, one host thread each, and exchange through device-to-device copies and a
//          host rendezvous — for single-process hosts and for rehearsing many ranks on few GPUs (RCCL refuses two
//          ranks on one device).
// Every collective of a communicator is issued on the ENGINE's stream, in program order: stream-ordered with the
// kernels around it, no second stream and no event hand-overs.
// No counterpart in the reference (it has no multi-device code).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "kernels.h"

struct slam_comm;

namespace slam {

int comm_rank(const slam_comm* c);
int comm_world(const slam_comm* c);
slam_engine* comm_engine(const slam_comm* c);

// All of these return a slam_status.  Buffers are device memory of the communicator's engine.
// in-place MAX over ranks of `count` floats; the engine's stream sees the result
int comm_all_reduce_max_f32(slam_comm* c, float* d_buf, int count);
// d_recv[q * bytes .. ) = rank q's d_send[0 .. bytes)
int comm_all_gather(slam_comm* c, const void* d_send, void* d_recv, size_t bytes);
// two all-gathers as one grouped operation (one launch, one latency)
int comm_all_gather2(slam_comm* c, const void* d_send_a, void* d_recv_a, size_t bytes_a, const void* d_send_b,
                     void* d_recv_b, size_t bytes_b);
// begin / finish form: the result may be used only after comm_all_gather_finish (today both are stream-ordered, so
// finish has nothing to wait for; the pair marks where a transport with overlap would hand over)
int comm_all_gather_begin(slam_comm* c, const void* d_send, void* d_recv, size_t bytes);
int comm_all_gather_finish(slam_comm* c);
// rank q receives send_floats[q] floats from my d_send (blocks in rank order); I receive recv_floats[q] from q
int comm_all_to_all_f32(slam_comm* c, const float* d_send, const int64_t* send_floats, float* d_recv,
                        const int64_t* recv_floats);

// ---- failure path (RCCL with more than one rank has never run on hardware here: none of this has met a real failure)
// give up: marks the communicator dead, breaks an in-process group / aborts the RCCL communicator so that the peers
// do not wait for this rank for ever.  Every later call on the communicator returns SLAM_ERR_COMM.
int comm_abort(slam_comm* c);
// SLAM_OK, or SLAM_ERR_COMM when the communicator is dead, its group is broken or RCCL reports an asynchronous error
// (the communicator is then aborted)
int comm_poll(slam_comm* c);
// hipStreamSynchronize / a wait for a flag in mapped host memory that poll the communicator and give up (abort) after
// SLAM_COMM_TIMEOUT_S seconds (default 120)
int comm_wait_stream(slam_comm* c);
int comm_wait_flag(slam_comm* c, const volatile uint32_t* flag, uint32_t seq);

}  // namespace slam
