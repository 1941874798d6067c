// comm.hip — RCCL communicator behind the C ABI: the collectives of the data-parallel training step on the CALLER's HIP stream
// (bucketed gradient all-reduce, buffer broadcast, SyncBN statistics), one process per GPU over xGMI.
//
// Replaces torch.nn.parallel.DistributedDataParallel's reducer + ProcessGroupNCCL reached from reference trainer.py:312-313
// (DDP wrap) and src/utils/distributed.py:82-98 (init_process_group('nccl')): the reference's backend string is not kept —
// librccl is bound directly (dlopen: the copy already mapped into the process by the host framework if there is one, so the
// process keeps ONE HIP runtime; otherwise $CVHIP_RCCL_PATH, librccl.so.1, librccl.so), the rendezvous of the 128-byte unique
// id is the host side's business (any byte channel), and every collective is asynchronous, non-allocating and CAPTURABLE in a
// hipGraph (RCCL enqueues kernels on the given stream; no host synchronisation here).
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>

#include "common.h"

namespace cvhip {

struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclReduceScatter) ReduceScatter = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  std::string error;
};

static RcclApi g_rccl;
static std::once_flag g_rccl_once;

static void load_rccl() {
  RcclApi& a = g_rccl;
  const char* env = getenv("CVHIP_RCCL_PATH");
  const char* names[] = {env, "librccl.so", "librccl.so.1"};
  // 1) a copy that is already mapped (RTLD_NOLOAD): the host framework's bundled librccl shares the process's HIP runtime
  for (const char* n : names) {
    if (!n || !*n || a.handle) continue;
    a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  }
  // 2) otherwise load one
  for (const char* n : names) {
    if (!n || !*n || a.handle) continue;
    a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!a.handle) {
    const char* e = dlerror();
    a.error = std::string("librccl not found (set CVHIP_RCCL_PATH): ") + (e ? e : "");
    return;
  }
#define CVHIP_SYM(field, name)                                              \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, name));     \
  if (!a.field) {                                                           \
    a.error = std::string("librccl lacks symbol ") + name;                  \
    return;                                                                 \
  }
  CVHIP_SYM(GetUniqueId, "ncclGetUniqueId")
  CVHIP_SYM(CommInitRank, "ncclCommInitRank")
  CVHIP_SYM(CommDestroy, "ncclCommDestroy")
  CVHIP_SYM(CommAbort, "ncclCommAbort")
  CVHIP_SYM(AllReduce, "ncclAllReduce")
  CVHIP_SYM(Broadcast, "ncclBroadcast")
  CVHIP_SYM(ReduceScatter, "ncclReduceScatter")
  CVHIP_SYM(AllGather, "ncclAllGather")
  CVHIP_SYM(GroupStart, "ncclGroupStart")
  CVHIP_SYM(GroupEnd, "ncclGroupEnd")
  CVHIP_SYM(GetErrorString, "ncclGetErrorString")
  CVHIP_SYM(GetVersion, "ncclGetVersion")
#undef CVHIP_SYM
}

static RcclApi* rccl() {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.error.empty()) {
    set_last_error(g_rccl.error.c_str(), hipErrorSharedObjectInitFailed);
    return nullptr;
  }
  return &g_rccl;
}

struct Comm {
  ncclComm_t comm;
  int world, rank;
};

static int fail(RcclApi* a, const char* what, ncclResult_t r) {
  std::string msg = std::string(what) + ": " + (a && a->GetErrorString ? a->GetErrorString(r) : "rccl error");
  set_last_error(msg.c_str(), hipErrorUnknown);
  return CVHIP_ERR_LAUNCH;
}

static bool dtype_of(int32_t code, ncclDataType_t* dt) {
  switch (code) {
    case CVHIP_DTYPE_F32: *dt = ncclFloat32; return true;
    case CVHIP_DTYPE_F64: *dt = ncclFloat64; return true;
    case CVHIP_DTYPE_I32: *dt = ncclInt32; return true;
    case CVHIP_DTYPE_BF16: *dt = ncclBfloat16; return true;
    case CVHIP_DTYPE_U8: *dt = ncclUint8; return true;
    default: return false;
  }
}

static bool op_of(int32_t code, ncclRedOp_t* op) {
  switch (code) {
    case CVHIP_RED_SUM: *op = ncclSum; return true;
    case CVHIP_RED_MAX: *op = ncclMax; return true;
    case CVHIP_RED_MIN: *op = ncclMin; return true;
    default: return false;
  }
}

}  // namespace cvhip

using namespace cvhip;

extern "C" {

int cvhip_comm_available(void) { return rccl() ? 1 : 0; }

int cvhip_comm_rccl_version(void) {
  RcclApi* a = rccl();
  if (!a) return 0;
  int v = 0;
  return a->GetVersion(&v) == ncclSuccess ? v : 0;
}

int cvhip_comm_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

int cvhip_comm_get_unique_id(void* id_out) {
  if (!id_out) return CVHIP_ERR_INVALID;
  RcclApi* a = rccl();
  if (!a) return CVHIP_ERR_UNSUPPORTED;
  ncclUniqueId id;
  ncclResult_t r = a->GetUniqueId(&id);
  if (r != ncclSuccess) return fail(a, "ncclGetUniqueId", r);
  memcpy(id_out, &id, NCCL_UNIQUE_ID_BYTES);
  return CVHIP_OK;
}

int cvhip_comm_init_rank(void** comm_out, int32_t world, int32_t rank, const void* unique_id) {
  if (!comm_out || !unique_id || world < 1 || rank < 0 || rank >= world) return CVHIP_ERR_INVALID;
  RcclApi* a = rccl();
  if (!a) return CVHIP_ERR_UNSUPPORTED;
  ncclUniqueId id;
  memcpy(&id, unique_id, NCCL_UNIQUE_ID_BYTES);
  Comm* c = new Comm{nullptr, world, rank};
  ncclResult_t r = a->CommInitRank(&c->comm, world, id, rank);  // binds the calling thread's current HIP device
  if (r != ncclSuccess) {
    delete c;
    return fail(a, "ncclCommInitRank", r);
  }
  *comm_out = c;
  return CVHIP_OK;
}

int cvhip_comm_destroy(void* comm) {
  if (!comm) return CVHIP_OK;
  RcclApi* a = rccl();
  Comm* c = static_cast<Comm*>(comm);
  int st = CVHIP_OK;
  if (a && c->comm) {
    ncclResult_t r = a->CommDestroy(c->comm);
    if (r != ncclSuccess) st = fail(a, "ncclCommDestroy", r);
  }
  delete c;
  return st;
}

int cvhip_comm_world(void* comm) { return comm ? static_cast<Comm*>(comm)->world : 0; }
int cvhip_comm_rank(void* comm) { return comm ? static_cast<Comm*>(comm)->rank : -1; }

int cvhip_comm_allreduce(void* comm, void* buf, int64_t count, int32_t dtype, int32_t op, void* stream) {
  if (!comm || (!buf && count > 0) || count < 0) return CVHIP_ERR_INVALID;
  if (count == 0) return CVHIP_OK;
  RcclApi* a = rccl();
  if (!a) return CVHIP_ERR_UNSUPPORTED;
  ncclDataType_t dt;
  ncclRedOp_t rop;
  if (!dtype_of(dtype, &dt) || !op_of(op, &rop)) return CVHIP_ERR_INVALID;
  ncclResult_t r = a->AllReduce(buf, buf, (size_t)count, dt, rop, static_cast<Comm*>(comm)->comm, (hipStream_t)stream);
  return r == ncclSuccess ? CVHIP_OK : fail(a, "ncclAllReduce", r);
}

int cvhip_allreduce_bucket(void* comm, void* buf_f32, int64_t count, void* stream) {
  return cvhip_comm_allreduce(comm, buf_f32, count, CVHIP_DTYPE_F32, CVHIP_RED_SUM, stream);
}

int cvhip_comm_broadcast(void* comm, void* buf, int64_t bytes, int32_t root, void* stream) {
  if (!comm || (!buf && bytes > 0) || bytes < 0) return CVHIP_ERR_INVALID;
  Comm* c = static_cast<Comm*>(comm);
  if (root < 0 || root >= c->world) return CVHIP_ERR_INVALID;
  if (bytes == 0) return CVHIP_OK;
  RcclApi* a = rccl();
  if (!a) return CVHIP_ERR_UNSUPPORTED;
  ncclResult_t r = a->Broadcast(buf, buf, (size_t)bytes, ncclUint8, root, c->comm, (hipStream_t)stream);
  return r == ncclSuccess ? CVHIP_OK : fail(a, "ncclBroadcast", r);
}

// In-place reduce-scatter + all-gather of a gradient range: rank r ends up with the complete sum in
// buf[r*chunk, (r+1)*chunk) after the first phase and everybody with everything after the second; `count` must be a
// multiple of the world size. Same result as cvhip_allreduce_bucket; lets the caller put the optimizer's shard between the
// two phases (ZeRO-1 style) or overlap the all-gather with the next forward.
int cvhip_comm_reduce_scatter_f32(void* comm, void* buf_f32, int64_t count, void* stream) {
  if (!comm || !buf_f32 || count < 0) return CVHIP_ERR_INVALID;
  Comm* c = static_cast<Comm*>(comm);
  if (count % c->world) return CVHIP_ERR_INVALID;
  if (count == 0) return CVHIP_OK;
  RcclApi* a = rccl();
  if (!a) return CVHIP_ERR_UNSUPPORTED;
  const size_t chunk = (size_t)(count / c->world);
  float* base = static_cast<float*>(buf_f32);
  ncclResult_t r = a->ReduceScatter(base, base + chunk * c->rank, chunk, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream);
  return r == ncclSuccess ? CVHIP_OK : fail(a, "ncclReduceScatter", r);
}

int cvhip_comm_all_gather_f32(void* comm, void* buf_f32, int64_t count, void* stream) {
  if (!comm || !buf_f32 || count < 0) return CVHIP_ERR_INVALID;
  Comm* c = static_cast<Comm*>(comm);
  if (count % c->world) return CVHIP_ERR_INVALID;
  if (count == 0) return CVHIP_OK;
  RcclApi* a = rccl();
  if (!a) return CVHIP_ERR_UNSUPPORTED;
  const size_t chunk = (size_t)(count / c->world);
  float* base = static_cast<float*>(buf_f32);
  ncclResult_t r = a->AllGather(base + chunk * c->rank, base, chunk, ncclFloat32, c->comm, (hipStream_t)stream);
  return r == ncclSuccess ? CVHIP_OK : fail(a, "ncclAllGather", r);
}

}  // extern "C"
