// RCCL collectives behind the C ABI: the exchanges of the row-wise sharded tables and the all-reduce of
// replicated dense gradients, for hosts that drive the library without torch.distributed.
//
// They replace the reference's parameter-server traffic (recommendation/code/model_trainer.py:124-136:
// every variable read and optimizer apply is a gRPC hop to the ps task; recommendation/train_job.py:41-42
// starts the workers): one process per GPU, one communicator per process, fixed-size exchanges enqueued on
// the caller's stream, nothing synchronises the host.
//
// RCCL is resolved at run time (dlopen of librccl.so, reusing a copy the process has already loaded - a
// PyTorch process has its own): the library itself links against the HIP runtime only and loads on a
// machine without RCCL; there the calls below fail with MI_ERR_UNSUPPORTED.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace {

// the slice of rccl.h this file uses (types are ABI-stable: an opaque pointer, a 128-byte id, small enums)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;   // ncclSuccess == 0
enum { NCCL_INT8 = 0, NCCL_INT32 = 2, NCCL_FLOAT32 = 7 };
enum { NCCL_SUM = 0 };

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*AllToAll)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  bool ok = false;
};

Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
  const char* names[] = {"librccl.so", "librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {  // a copy already in the process first (RTLD_NOLOAD), then the system's
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (h) break;
  }
  if (!h) {
    for (const char* n : names) {
      h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
  }
  if (!h) return;
  Rccl& r = g_rccl;
  r.handle = h;
#define SYM(field, name) *(void**) (&r.field) = dlsym(h, name)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(CommCount, "ncclCommCount");
  SYM(CommUserRank, "ncclCommUserRank");
  SYM(GetErrorString, "ncclGetErrorString");
  SYM(AllToAll, "ncclAllToAll");
  SYM(AllGather, "ncclAllGather");
  SYM(AllReduce, "ncclAllReduce");
#undef SYM
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommCount && r.CommUserRank && r.GetErrorString &&
         r.AllToAll && r.AllGather && r.AllReduce;
}

int need_rccl(const char* who) {
  std::call_once(g_once, load_rccl);
  if (!g_rccl.ok) {
    mi_set_error("%s: librccl.so could not be loaded (%s)", who, g_rccl.handle ? "a symbol is missing" : "dlopen failed");
    return MI_ERR_UNSUPPORTED;
  }
  return MI_OK;
}

#define MI_RCCL(call, who)                                                         \
  do {                                                                             \
    ncclResult_t r_ = (call);                                                      \
    if (r_ != 0) {                                                                 \
      mi_set_error("%s: %s failed: %s", who, #call, g_rccl.GetErrorString(r_));    \
      return MI_ERR_HIP;                                                           \
    }                                                                              \
  } while (0)

struct Comm {
  ncclComm_t c;
  int n_ranks, rank;
};

int check_comm(const void* comm, const char* who) {
  int rc = need_rccl(who);
  if (rc) return rc;
  MI_REQUIRE(comm, "%s: null communicator", who);
  return MI_OK;
}

}  // namespace

extern "C" int mi_comm_unique_id(void* id_out) {
  const char* who = "mi_comm_unique_id";
  int rc = need_rccl(who);
  if (rc) return rc;
  MI_REQUIRE(id_out, "%s: null pointer", who);
  ncclUniqueId id;
  MI_RCCL(g_rccl.GetUniqueId(&id), who);
  memcpy(id_out, &id, sizeof(id));
  return MI_OK;
}

extern "C" int mi_comm_init(const void* unique_id, int n_ranks, int rank, void** comm_out) {
  const char* who = "mi_comm_init";
  int rc = need_rccl(who);
  if (rc) return rc;
  MI_REQUIRE(unique_id && comm_out, "%s: null pointer", who);
  MI_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "%s: rank %d of %d", who, rank, n_ranks);
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t c = nullptr;
  MI_RCCL(g_rccl.CommInitRank(&c, n_ranks, id, rank), who);
  Comm* out = new Comm{c, n_ranks, rank};
  *comm_out = out;
  return MI_OK;
}

extern "C" int mi_comm_destroy(void* comm) {
  const char* who = "mi_comm_destroy";
  if (!comm) return MI_OK;
  int rc = need_rccl(who);
  if (rc) return rc;
  Comm* c = static_cast<Comm*>(comm);
  MI_RCCL(g_rccl.CommDestroy(c->c), who);
  delete c;
  return MI_OK;
}

extern "C" int mi_comm_size(const void* comm, int* n_ranks, int* rank) {
  int rc = check_comm(comm, "mi_comm_size");
  if (rc) return rc;
  const Comm* c = static_cast<const Comm*>(comm);
  if (n_ranks) *n_ranks = c->n_ranks;
  if (rank) *rank = c->rank;
  return MI_OK;
}

extern "C" int mi_a2a_indices(void* comm, const int32_t* send, int32_t* recv, int64_t ints_per_pair, void* stream) {
  const char* who = "mi_a2a_indices";
  int rc = check_comm(comm, who);
  if (rc) return rc;
  MI_REQUIRE(send && recv && send != recv && ints_per_pair >= 0, "%s: bad buffers / count", who);
  if (ints_per_pair == 0) return MI_OK;
  Comm* c = static_cast<Comm*>(comm);
  MI_RCCL(g_rccl.AllToAll(send, recv, (size_t) ints_per_pair, NCCL_INT32, c->c, (hipStream_t) stream), who);
  return MI_OK;
}

extern "C" int mi_a2a_pooled(void* comm, const float* send, float* recv, int64_t floats_per_pair, void* stream) {
  const char* who = "mi_a2a_pooled";
  int rc = check_comm(comm, who);
  if (rc) return rc;
  MI_REQUIRE(send && recv && send != recv && floats_per_pair >= 0, "%s: bad buffers / count", who);
  if (floats_per_pair == 0) return MI_OK;
  Comm* c = static_cast<Comm*>(comm);
  MI_RCCL(g_rccl.AllToAll(send, recv, (size_t) floats_per_pair, NCCL_FLOAT32, c->c, (hipStream_t) stream), who);
  return MI_OK;
}

extern "C" int mi_allgather_bytes(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream) {
  const char* who = "mi_allgather_bytes";
  int rc = check_comm(comm, who);
  if (rc) return rc;
  MI_REQUIRE(send && recv && bytes_per_rank >= 0, "%s: bad buffers / count", who);
  if (bytes_per_rank == 0) return MI_OK;
  Comm* c = static_cast<Comm*>(comm);
  MI_RCCL(g_rccl.AllGather(send, recv, (size_t) bytes_per_rank, NCCL_INT8, c->c, (hipStream_t) stream), who);
  return MI_OK;
}

extern "C" int mi_allreduce_dense(void* comm, float* buf, int64_t n, void* stream) {
  const char* who = "mi_allreduce_dense";
  int rc = check_comm(comm, who);
  if (rc) return rc;
  MI_REQUIRE(buf && n >= 0, "%s: bad buffer / count", who);
  if (n == 0) return MI_OK;
  Comm* c = static_cast<Comm*>(comm);
  MI_RCCL(g_rccl.AllReduce(buf, buf, (size_t) n, NCCL_FLOAT32, NCCL_SUM, c->c, (hipStream_t) stream), who);
  return MI_OK;
}
