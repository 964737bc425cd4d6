// Gradient exchange of the data-parallel path on RCCL (reference: DistributedDataParallel over NCCL, trainer.py:212-219,297).
// One communicator per process (= per GPU); the exchange is ONE in-place fp32 sum all-reduce of the flat gradient buffer
// per step over xGMI, plus a one-time parameter broadcast.  RCCL is resolved at run time from the copy already loaded in
// the process (PyTorch-ROCm ships librccl.so; `torch.distributed`'s "nccl" backend is that same library), so the .so has
// no link-time dependency on it and two RCCL copies never coexist.
#include "common.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

namespace {
struct Api {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  bool ok;
};
Api* api() {
  static Api a = [] {
    Api x{};
    // 1. symbols of an RCCL the process ALREADY has, under whatever file name it was loaded (torch bundles its own copy; a
    //    hashed or versioned soname would defeat a dlopen(name, RTLD_NOLOAD) probe): the global symbol scope
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclAllReduce")) {
      // 2. loaded, but with local visibility (dlopen'ed by torch): probe the known names without loading anything
      h = nullptr;
      for (const char* name : {"librccl.so", "librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        if (h) break;
      }
      // 3. no RCCL in the process at all: load the system one.  (Never reached next to PyTorch-ROCm; a second copy beside
      //    torch's would have its own communicator state.)
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) return x;
    }
    x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(h, "ncclAllReduce"));
    x.Broadcast = reinterpret_cast<decltype(x.Broadcast)>(dlsym(h, "ncclBroadcast"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    x.ok = x.GetUniqueId && x.CommInitRank && x.AllReduce && x.Broadcast && x.CommDestroy;
    return x;
  }();
  return &a;
}
}  // namespace

struct nnr_dp_ctx { ncclComm_t comm; int rank, world; };

extern "C" int nnr_dp_unique_id(void* out128) {
  static_assert(sizeof(ncclUniqueId) == 128, "the boundary carries the id as 128 opaque bytes");
  if (!out128) return NNR_ERR_ARG;
  if (!api()->ok) return NNR_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (api()->GetUniqueId(&id) != ncclSuccess) return NNR_ERR_LAUNCH;
  memcpy(out128, &id, sizeof(id));
  return NNR_OK;
}

extern "C" int nnr_dp_init(const void* uid128, int rank, int world, nnr_dp_ctx** ctx) {
  if (!uid128 || !ctx || world < 1 || rank < 0 || rank >= world) return NNR_ERR_ARG;
  if (!api()->ok) return NNR_ERR_UNSUPPORTED;
  ncclUniqueId id;
  memcpy(&id, uid128, sizeof(id));
  ncclComm_t comm;
  if (api()->CommInitRank(&comm, world, id, rank) != ncclSuccess) return NNR_ERR_LAUNCH;     // binds to the CURRENT HIP device
  *ctx = new nnr_dp_ctx{comm, rank, world};
  return NNR_OK;
}

extern "C" int nnr_dp_allreduce(nnr_dp_ctx* ctx, float* flat, size_t n, hipStream_t stream) {
  if (!ctx || !flat) return NNR_ERR_ARG;
  if (n == 0) return NNR_OK;
  return api()->AllReduce(flat, flat, n, ncclFloat32, ncclSum, ctx->comm, stream) == ncclSuccess ? NNR_OK : NNR_ERR_LAUNCH;
}

extern "C" int nnr_dp_broadcast(nnr_dp_ctx* ctx, float* flat, size_t n, int root, hipStream_t stream) {
  if (!ctx || !flat || root < 0 || root >= ctx->world) return NNR_ERR_ARG;
  if (n == 0) return NNR_OK;
  return api()->Broadcast(flat, flat, n, ncclFloat32, root, ctx->comm, stream) == ncclSuccess ? NNR_OK : NNR_ERR_LAUNCH;
}

extern "C" int nnr_dp_destroy(nnr_dp_ctx* ctx) {
  if (!ctx) return NNR_OK;
  const bool ok = api()->CommDestroy(ctx->comm) == ncclSuccess;
  delete ctx;
  return ok ? NNR_OK : NNR_ERR_LAUNCH;
}
