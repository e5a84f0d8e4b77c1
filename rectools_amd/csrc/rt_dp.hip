// Data-parallel gradient exchange over RCCL behind the C ABI (rt_dp_init / allreduce / finalize).
//
// The reference delegates the exchange to Lightning's DDP (`transformers/base.py:367-380`): one all-reduce of the
// gradients per step over NCCL.  Here the flat fp32 gradient buffer of `FlatAdam` is summed in place with ONE
// ncclAllReduce on the caller's stream (DDP's 1/world is folded into the Adam kernel).  RCCL is resolved at RUN time
// with dlopen: a process that imported torch already holds torch's copy of librccl (torch.distributed's "nccl" backend IS
// RCCL on ROCm), and linking a second copy at build time would put two RCCL instances into one process — the same
// reason `_lib.load()` lets this library bind to torch's HIP runtime.  No RCCL symbol is referenced at link time; the
// header is used for its types only.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <rccl/rccl.h>

#include "rt_common.h"

namespace {

struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclReduceScatter) ReduceScatter = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool tried = false;
};
RcclApi g_rccl;
thread_local char g_dp_error[256] = "";

bool load_rccl() {
  if (g_rccl.tried) return g_rccl.handle != nullptr;
  g_rccl.tried = true;
  const char* env = getenv("RT_RCCL_LIB");
  const char* names[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (int pass = 0; pass < 2 && !h; ++pass)      // pass 0: only a copy that is already in the process (torch's)
    for (const char* n : names) {
      if (!n || !*n) continue;
      h = dlopen(n, RTLD_LAZY | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (h) break;
    }
  if (!h) { snprintf(g_dp_error, sizeof(g_dp_error), "librccl.so not found (set RT_RCCL_LIB)"); return false; }
  g_rccl.GetUniqueId = reinterpret_cast<decltype(&ncclGetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(&ncclCommInitRank)>(dlsym(h, "ncclCommInitRank"));
  g_rccl.AllReduce = reinterpret_cast<decltype(&ncclAllReduce)>(dlsym(h, "ncclAllReduce"));
  g_rccl.Broadcast = reinterpret_cast<decltype(&ncclBroadcast)>(dlsym(h, "ncclBroadcast"));
  g_rccl.ReduceScatter = reinterpret_cast<decltype(&ncclReduceScatter)>(dlsym(h, "ncclReduceScatter"));
  g_rccl.AllGather = reinterpret_cast<decltype(&ncclAllGather)>(dlsym(h, "ncclAllGather"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(&ncclCommDestroy)>(dlsym(h, "ncclCommDestroy"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(&ncclGetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.Broadcast || !g_rccl.CommDestroy) {
    snprintf(g_dp_error, sizeof(g_dp_error), "librccl.so lacks an expected symbol");
    dlclose(h);
    return false;
  }
  g_rccl.handle = h;
  return true;
}

int nccl_status(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return RT_OK;
  snprintf(g_dp_error, sizeof(g_dp_error), "%s: %s (%d)", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error", (int)r);
  return RT_ERR_LAUNCH;
}

}  // namespace

extern "C" {

const char* rt_dp_last_error(void) { return g_dp_error; }

// 128 opaque bytes created by ONE rank (rank 0) and handed to every rank out of band (torch.distributed store, MPI, a file).
int rt_dp_unique_id(void* out128) {
  if (out128 == nullptr) return RT_ERR_INVALID_ARG;
  if (!load_rccl()) return RT_ERR_UNSUPPORTED;
  ncclUniqueId id;
  const int rc = nccl_status(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  if (rc == RT_OK) memcpy(out128, &id, sizeof(id));
  return rc;
}

// Collective over all `world` ranks (one process per GPU, the current HIP device is this rank's GPU).
int rt_dp_init(const void* uid128, int32_t rank, int32_t world, void** comm_out) {
  if (uid128 == nullptr || comm_out == nullptr || world < 1 || rank < 0 || rank >= world) return RT_ERR_INVALID_ARG;
  if (!load_rccl()) return RT_ERR_UNSUPPORTED;
  ncclUniqueId id;
  memcpy(&id, uid128, sizeof(id));
  ncclComm_t comm = nullptr;
  const int rc = nccl_status(g_rccl.CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
  *comm_out = rc == RT_OK ? comm : nullptr;
  return rc;
}

// buf[0:n] <- sum over ranks, in place, fp32, asynchronous on `stream` (the stream the gradient kernels ran on).
int rt_dp_allreduce(void* comm, float* buf, int64_t n, hipStream_t stream) {
  if (comm == nullptr || (buf == nullptr && n > 0) || n < 0) return RT_ERR_INVALID_ARG;
  if (n == 0) return RT_OK;
  if (!load_rccl()) return RT_ERR_UNSUPPORTED;
  return nccl_status(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, reinterpret_cast<ncclComm_t>(comm), stream), "ncclAllReduce");
}

// buf[0:n] of rank `root` to every rank (parameters and Adam moments at the start of a data-parallel fit).
int rt_dp_broadcast(void* comm, float* buf, int64_t n, int32_t root, hipStream_t stream) {
  if (comm == nullptr || (buf == nullptr && n > 0) || n < 0) return RT_ERR_INVALID_ARG;
  if (n == 0) return RT_OK;
  if (!load_rccl()) return RT_ERR_UNSUPPORTED;
  return nccl_status(g_rccl.Broadcast(buf, buf, (size_t)n, ncclFloat32, root, reinterpret_cast<ncclComm_t>(comm), stream), "ncclBroadcast");
}

// The exchange for table-dominated models (C4 1 GB, C5-train 10 GB of gradient per step): every rank keeps only ITS
// 1/world slice of the reduced gradient, runs Adam on that slice of (p, m, v) — the moments never leave their shard — and the updated
// parameter slices are gathered back.  Same bytes on the links as the all-reduce, 1/world of the optimiser pass per GPU.
// rt_dp_reduce_scatter: recv[0:n] <- sum over ranks of send[rank * n : (rank + 1) * n]  (send holds world * n floats).
int rt_dp_reduce_scatter(void* comm, const float* send, float* recv, int64_t n, hipStream_t stream) {
  if (comm == nullptr || n < 0 || (n > 0 && (send == nullptr || recv == nullptr))) return RT_ERR_INVALID_ARG;
  if (n == 0) return RT_OK;
  if (!load_rccl() || !g_rccl.ReduceScatter) return RT_ERR_UNSUPPORTED;
  return nccl_status(g_rccl.ReduceScatter(send, recv, (size_t)n, ncclFloat32, ncclSum, reinterpret_cast<ncclComm_t>(comm), stream),
                     "ncclReduceScatter");
}
// rt_dp_allgather: recv[r * n : (r + 1) * n] <- rank r's send[0:n]  (recv holds world * n floats; send may be the rank's own slice of recv).
int rt_dp_allgather(void* comm, const float* send, float* recv, int64_t n, hipStream_t stream) {
  if (comm == nullptr || n < 0 || (n > 0 && (send == nullptr || recv == nullptr))) return RT_ERR_INVALID_ARG;
  if (n == 0) return RT_OK;
  if (!load_rccl() || !g_rccl.AllGather) return RT_ERR_UNSUPPORTED;
  return nccl_status(g_rccl.AllGather(send, recv, (size_t)n, ncclFloat32, reinterpret_cast<ncclComm_t>(comm), stream), "ncclAllGather");
}

int rt_dp_finalize(void* comm) {
  if (comm == nullptr) return RT_OK;
  if (!load_rccl()) return RT_ERR_UNSUPPORTED;
  return nccl_status(g_rccl.CommDestroy(reinterpret_cast<ncclComm_t>(comm)), "ncclCommDestroy");
}

}  // extern "C"
