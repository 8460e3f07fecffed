// comm.hpp — the exchange transport of the sharded group-by (comm.hip): RCCL point-to-point on the library's stream,
// or an all-to-all the host supplies (a host that owns its own communicator, and the one-GPU rehearsals of the tests).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/plhip.h"

struct plhip_comm {
  int world = 1, rank = 0, device = 0;
  void *nccl = nullptr; // ncclComm_t (RCCL transport)
  plhip_alltoall_fn a2a = nullptr; // host transport
  plhip_alltoallv_fn a2av = nullptr;
  void *user = nullptr;
};

// Equal split: peer p gets send[p * bytes, (p + 1) * bytes) and delivers into recv[p * bytes, ...).  RCCL: enqueued on `s`
// (ncclGroupStart; ncclSend / ncclRecv per peer; ncclGroupEnd).  Host transport: `s` is drained, then the callback runs.
int comm_all_to_all(plhip_comm *c, const void *send, void *recv, size_t bytes, hipStream_t s, std::string &err);
// Variable split: byte counts and byte offsets per peer (offsets are the prefix sums of the counts).
int comm_all_to_all_v(plhip_comm *c, const void *send, const int64_t *send_bytes, void *recv, const int64_t *recv_bytes, hipStream_t s,
                      std::string &err);
int comm_unique_id(uint8_t *id128, std::string &err);
int comm_init_rccl(const uint8_t *id128, int world, int rank, int device, plhip_comm **out, std::string &err);
// world / rank as the transport reports them (RCCL: ncclCommCount / ncclCommUserRank); transport: 1 = RCCL, 0 = host callbacks
int comm_info(plhip_comm *c, int *world, int *rank, int *transport, std::string &err);
void comm_free(plhip_comm *c);
