// comm.hip — the data-parallel exchange of the search loop inside the C-ABI (`vstar_allgather_results`).
//
// One process per GPU, weights replicated, each engine step's crops dealt round-robin over the ranks; after the step every rank
// needs ALL records to take the same best-first decision (visual_search.py:399-478 semantics).  This file owns that one collective:
// an RCCL all-gather of the fixed-size `vstar_result` records (193,568 B each) on the ENGINE'S OWN STREAM, so that it queues behind
// the kernels that produce the records and overlaps with whatever the host does next — no torch.distributed, no host bounce.
//
// RCCL is bound at RUN time (dlopen of librccl.so.1, the copy already in the process when PyTorch-ROCm is loaded, else the system
// one): libvstar_hip.so has no link-time dependency on it, and a host that never calls vstar_comm_* never loads it.
// xGMI is point-to-point (7 links x ~153 GB/s per GPU): 8 ranks x 32 records = 6.2 MB per rank per step is latency-bound (ring
// all-gather, 7 hops), which is why the record is one flat fp32 block and the gather ONE call per step.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstring>
#include <string>
#include "../../include/vstar_hip.h"


// accessors implemented in engine.hip (the handle's layout is private to that file)
int vstar_handle_device(vstar_handle* h);
void vstar_handle_set_error(vstar_handle* h, const char* msg);
void** vstar_handle_comm_slot(vstar_handle* h);      // where the engine keeps its communicator (void* ncclComm_t), null-initialised

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

Rccl& rccl() {
  static Rccl r;
  if (r.lib || !r.err.empty()) return r;
  // RTLD_NOLOAD first: reuse the RCCL that torch (or the host) already mapped — two copies of the library in one process would
  // each want to own the xGMI topology
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (r.lib) break;
  }
  if (!r.lib)
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
  if (!r.lib) { r.err = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : ""); return r; }
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
  r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) { r.err = "RCCL symbols missing"; r.lib = nullptr; }
  return r;
}

int fail(vstar_handle* h, const char* what, ncclResult_t rc) {
  Rccl& r = rccl();
  std::string m = std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
  vstar_handle_set_error(h, m.c_str());
  return VSTAR_ERR_HIP;
}

}  // namespace

extern "C" {

int vstar_comm_unique_id(uint8_t* id_out) {
  if (!id_out) return VSTAR_ERR_INVALID;
  Rccl& r = rccl();
  if (!r.lib) return VSTAR_ERR_STATE;
  ncclUniqueId id;
  if (r.GetUniqueId(&id) != ncclSuccess) return VSTAR_ERR_HIP;
  static_assert(sizeof(id) == VSTAR_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id_out, &id, sizeof(id));
  return VSTAR_OK;
}

int vstar_comm_init(vstar_handle* h, const uint8_t* id, int world, int rank) {
  if (!h || !id || world < 1 || rank < 0 || rank >= world) return VSTAR_ERR_INVALID;
  Rccl& r = rccl();
  if (!r.lib) { vstar_handle_set_error(h, r.err.c_str()); return VSTAR_ERR_STATE; }
  void** slot = vstar_handle_comm_slot(h);
  if (*slot) { vstar_handle_set_error(h, "communicator already initialised"); return VSTAR_ERR_STATE; }
  hipSetDevice(vstar_handle_device(h));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm = nullptr;
  const ncclResult_t rc = r.CommInitRank(&comm, world, uid, rank);
  if (rc != ncclSuccess) return fail(h, "ncclCommInitRank", rc);
  *slot = (void*)comm;
  return VSTAR_OK;
}

int vstar_allgather_results(vstar_handle* h, const vstar_result* local_dev, int n_local, vstar_result* gathered_dev, unsigned flags) {
  if (!h || !local_dev || !gathered_dev || n_local <= 0) return VSTAR_ERR_INVALID;
  Rccl& r = rccl();
  void** slot = vstar_handle_comm_slot(h);
  if (!r.lib || !*slot) { vstar_handle_set_error(h, "vstar_comm_init has not been called"); return VSTAR_ERR_STATE; }
  hipSetDevice(vstar_handle_device(h));
  void* st = vstar_stream(h);
  const size_t count = (size_t)n_local * (sizeof(vstar_result) / sizeof(float));
  const ncclResult_t rc = r.AllGather(local_dev, gathered_dev, count, ncclFloat, (ncclComm_t)*slot, (hipStream_t)st);
  if (rc != ncclSuccess) return fail(h, "ncclAllGather", rc);
  if (!(flags & VSTAR_F_NO_SYNC) && hipStreamSynchronize((hipStream_t)st) != hipSuccess) {
    vstar_handle_set_error(h, "stream sync failed after ncclAllGather");
    return VSTAR_ERR_HIP;
  }
  return VSTAR_OK;
}

int vstar_comm_destroy(vstar_handle* h) {
  if (!h) return VSTAR_ERR_INVALID;
  void** slot = vstar_handle_comm_slot(h);
  if (*slot) {
    Rccl& r = rccl();
    if (r.lib) r.CommDestroy((ncclComm_t)*slot);
    *slot = nullptr;
  }
  return VSTAR_OK;
}

}  // extern "C"
