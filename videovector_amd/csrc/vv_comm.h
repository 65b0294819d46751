// vv_comm.h -- internal interface of comm.hip (the data-parallel gradient exchange behind vv_comm_* in include/videovec.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string>

#include "../../include/videovec.h"

namespace vv {
struct Comm;
Comm* comm_create(int world, int rank, const char* id_path, int transport, size_t n_floats, std::string* err);
void comm_destroy(Comm* c);
// all-reduce(sum) of buf[off .. off+n) in place on the communication stream, started after `after` (an event of the
// compute stream, or null); comm_done_event marks its completion
int comm_allreduce(Comm* c, float* buf, size_t off, size_t n, hipEvent_t after);
int comm_allreduce_inline(Comm* c, float* buf, size_t n, hipStream_t stream);   // on the caller's stream (RCCL); 1 = not available
// Sharded update (api.hip: the `sharded` schedule), both in place on the communication stream:
//   comm_reduce_scatter  buf holds `world` shards of shard_floats each; on return rank r's shard holds the sum over the ranks of that
//                        shard (the other shards are unspecified); started after `after`
//   comm_allgather       n buffers at once (one grouped launch on RCCL), buffer i = `world` shards of shard_bytes[i]; rank r's shard is
//                        the input, on return every shard holds its owner's bytes.  fence_after (direct peer transport): the ranks meet
//                        once more behind the gather -- for a gather that no later collective orders against the owners' next writes
int comm_reduce_scatter(Comm* c, float* buf, size_t shard_floats, hipEvent_t after);
int comm_allgather(Comm* c, void* const* bufs, const size_t* shard_bytes, int n, int fence_after = 0);
hipEvent_t comm_done_event(Comm* c);
hipStream_t comm_stream(Comm* c);          // the communication stream (the overlapped update queues its kernels there)
void comm_use_stream(Comm* c, hipStream_t s);   // the collectives called from now on are queued on s (the caller's stream: no second stream, no
                                                // events between the two); nullptr: back to the communication stream
int comm_record_done(Comm* c);             // records comm_done_event behind everything queued on the communication stream
const char* comm_error(Comm* c);
const uint32_t* comm_fail_flag(Comm* c);   // device word, != 0 once an exchange gave up (direct peer transport; null otherwise): the update behind it is skipped
bool comm_failed(Comm* c);                 // a wait inside the exchange gave up (direct peer transport: a rank is missing); comm_error says so
int comm_world(Comm* c);
int comm_rank(Comm* c);
}  // namespace vv
