// comm.hip — exchange transport of plhip_group_by_agg_sharded.
//
// The reference's partitioned group-by hands every thread's partial tables to the thread that owns the hash partition
// (crates/polars-lazy/src/physical_plan/executors/group_by_partitioned.rs:320-360: concat of the per-thread frames, then
// the final group-by).  Between GPUs that hand-over is one all-to-all of packed records.  xGMI is point to point, so
// the all-to-all is exactly n - 1 sends and n - 1 receives per rank: ncclGroupStart; ncclSend / ncclRecv; ncclGroupEnd
// on the library's own stream, no staging, no extra copy.  RCCL is loaded at run time (dlopen): hosts that never shard
// do not need it, and a process that already holds an RCCL (a PyTorch host) shares that instance.
#include "comm.hpp"

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

namespace {

struct Rccl {
  void *h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  std::string why;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
  Rccl &R = g_rccl;
  const char *env = getenv("PLHIP_RCCL_LIBRARY");
  // an instance the process already holds first (RTLD_NOLOAD), then the usual places; PLHIP_RCCL_LIBRARY, when set, is the ONLY
  // name tried (the host said which RCCL it wants: a missing file is an error, not a reason to pick another copy)
  const bool only_env = env && *env;
  const char *names[] = {env, only_env ? nullptr : "librccl.so.1", only_env ? nullptr : "librccl.so",
                         only_env ? nullptr : "/opt/rocm/lib/librccl.so.1"};
  for (int pass = 0; pass < 2 && !R.h; pass++)
    for (const char *n : names) {
      if (!n || !*n) continue;
      R.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (R.h) break;
    }
  if (!R.h) {
    const char *e = dlerror(); // ONE call: dlerror() hands the message over and clears it
    R.why = std::string("librccl.so not found (set PLHIP_RCCL_LIBRARY): ") + (e ? e : "");
    return;
  }
#define SYM(field, name)                                                  \
  R.field = (decltype(R.field))dlsym(R.h, name);                          \
  if (!R.field) { R.why = std::string("librccl.so lacks ") + name; R.h = nullptr; return; }
  SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
  SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
  SYM(GetErrorString, "ncclGetErrorString") SYM(CommCount, "ncclCommCount") SYM(CommUserRank, "ncclCommUserRank")
#undef SYM
}

Rccl *rccl(std::string &err) {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.h) { err = g_rccl.why; return nullptr; }
  return &g_rccl;
}

#define NCCL_TRY(R, expr)                                                                    \
  do {                                                                                       \
    ncclResult_t _r = (expr);                                                                \
    if (_r != ncclSuccess) { err = std::string(#expr) + ": " + (R)->GetErrorString(_r); return PLHIP_ERR_HIP; } \
  } while (0)

} // namespace

int comm_unique_id(uint8_t *id128, std::string &err) {
  Rccl *R = rccl(err);
  if (!R) return PLHIP_ERR_NO_DEVICE;
  static_assert(sizeof(ncclUniqueId) == PLHIP_COMM_ID_BYTES, "plhip.h states the size of an RCCL unique id");
  ncclUniqueId id;
  NCCL_TRY(R, R->GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return PLHIP_OK;
}

int comm_init_rccl(const uint8_t *id128, int world, int rank, int device, plhip_comm **out, std::string &err) {
  Rccl *R = rccl(err);
  if (!R) return PLHIP_ERR_NO_DEVICE;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  NCCL_TRY(R, R->CommInitRank(&comm, world, id, rank)); // binds the CURRENT device (the caller has set it)
  plhip_comm *c = new plhip_comm();
  c->world = world; c->rank = rank; c->device = device; c->nccl = comm;
  *out = c;
  return PLHIP_OK;
}

int comm_info(plhip_comm *c, int *world, int *rank, int *transport, std::string &err) {
  *world = c->world; *rank = c->rank; *transport = c->nccl ? 1 : 0;
  if (c->nccl) { // what RCCL itself says about the communicator, not what the caller passed in
    Rccl *R = rccl(err);
    if (!R) return PLHIP_ERR_NO_DEVICE;
    NCCL_TRY(R, R->CommCount((ncclComm_t)c->nccl, world));
    NCCL_TRY(R, R->CommUserRank((ncclComm_t)c->nccl, rank));
  }
  return PLHIP_OK;
}

void comm_free(plhip_comm *c) {
  if (!c) return;
  if (c->nccl && g_rccl.h) g_rccl.CommDestroy((ncclComm_t)c->nccl);
  delete c;
}

int comm_all_to_all(plhip_comm *c, const void *send, void *recv, size_t bytes, hipStream_t s, std::string &err) {
  if (c->nccl) {
    Rccl *R = rccl(err);
    if (!R) return PLHIP_ERR_NO_DEVICE;
    NCCL_TRY(R, R->GroupStart());
    for (int p = 0; p < c->world; p++) {
      ncclResult_t a = R->Send((const uint8_t *)send + (size_t)p * bytes, bytes, ncclUint8, p, (ncclComm_t)c->nccl, s);
      ncclResult_t b = a == ncclSuccess ? R->Recv((uint8_t *)recv + (size_t)p * bytes, bytes, ncclUint8, p, (ncclComm_t)c->nccl, s) : a;
      if (b != ncclSuccess) { R->GroupEnd(); err = std::string("ncclSend/ncclRecv: ") + R->GetErrorString(b); return PLHIP_ERR_HIP; }
    }
    NCCL_TRY(R, R->GroupEnd());
    return PLHIP_OK;
  }
  if (!c->a2a) { err = "the communicator has no transport"; return PLHIP_ERR_INVALID; }
  hipError_t e = hipStreamSynchronize(s); // the host's transport knows nothing of the library's stream
  if (e != hipSuccess) { err = std::string("hipStreamSynchronize: ") + hipGetErrorString(e); return PLHIP_ERR_HIP; }
  const int32_t rc = c->a2a(c->user, send, recv, bytes);
  if (rc) { err = "the host's all-to-all failed with status " + std::to_string(rc); return PLHIP_ERR_COMPUTE; }
  return PLHIP_OK;
}

int comm_all_to_all_v(plhip_comm *c, const void *send, const int64_t *send_bytes, void *recv, const int64_t *recv_bytes, hipStream_t s,
                      std::string &err) {
  if (c->nccl) {
    Rccl *R = rccl(err);
    if (!R) return PLHIP_ERR_NO_DEVICE;
    NCCL_TRY(R, R->GroupStart());
    size_t so = 0, ro = 0;
    for (int p = 0; p < c->world; p++) {
      ncclResult_t a = ncclSuccess;
      if (send_bytes[p]) a = R->Send((const uint8_t *)send + so, (size_t)send_bytes[p], ncclUint8, p, (ncclComm_t)c->nccl, s);
      if (a == ncclSuccess && recv_bytes[p]) a = R->Recv((uint8_t *)recv + ro, (size_t)recv_bytes[p], ncclUint8, p, (ncclComm_t)c->nccl, s);
      if (a != ncclSuccess) { R->GroupEnd(); err = std::string("ncclSend/ncclRecv: ") + R->GetErrorString(a); return PLHIP_ERR_HIP; }
      so += (size_t)send_bytes[p]; ro += (size_t)recv_bytes[p];
    }
    NCCL_TRY(R, R->GroupEnd());
    return PLHIP_OK;
  }
  if (!c->a2av) { err = "the communicator has no variable-size transport"; return PLHIP_ERR_INVALID; }
  hipError_t e = hipStreamSynchronize(s);
  if (e != hipSuccess) { err = std::string("hipStreamSynchronize: ") + hipGetErrorString(e); return PLHIP_ERR_HIP; }
  const int32_t rc = c->a2av(c->user, send, send_bytes, recv, recv_bytes);
  if (rc) { err = "the host's all-to-all failed with status " + std::to_string(rc); return PLHIP_ERR_COMPUTE; }
  return PLHIP_OK;
}
