// Data-parallel collectives of the G.pt hot path on RCCL over xGMI, behind the C ABI (dp_init / dp_destroy /
// dp_allreduce_grads / dp_allgather / dp_broadcast / dp_barrier), for callers that do not go through torch.distributed:
//   gradient exchange of DistributedDataParallel                      main.py:229-240   -> gpt_dp_allreduce_grads (replicated step)
//                                                                                         gpt_dp_reduce_scatter + gpt_dp_allgather (sharded step)
//   all_gather / rank0_to_all / synchronize of Gpt/distributed.py     :152-189          -> gpt_dp_allgather / gpt_dp_broadcast / gpt_dp_barrier
//   init_process_group("nccl", ...)                                   main.py:328-333   -> gpt_dp_unique_id + gpt_dp_init
// One communicator per process (one process per GPU, as the reference launches them).  librccl is loaded on first use (dlopen),
// so the rest of the library carries no dependency on it.  The Python host's default route stays torch.distributed ("nccl" = RCCL).
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;

ncclComm_t g_comm = nullptr;
int g_rank = 0, g_world = 1;

// the word gpt_dp_barrier all-reduces: part of the code object, addressed per device through hipGetSymbolAddress
static __device__ float g_barrier_word;

int load_rccl() {
  if (g_rccl.handle) return 0;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  GPT_REQUIRE(h, "gpt_dp: cannot load librccl.so.1: %s", dlerror());
#define SYM(field, name)                                                           \
  do {                                                                             \
    *(void**)(&g_rccl.field) = dlsym(h, name);                                     \
    GPT_REQUIRE(g_rccl.field, "gpt_dp: librccl lacks %s", name);                   \
  } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllReduce, "ncclAllReduce");
  SYM(ReduceScatter, "ncclReduceScatter");
  SYM(AllGather, "ncclAllGather");
  SYM(Broadcast, "ncclBroadcast");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl.handle = h;
  return 0;
}

int rccl_status(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return 0;
  gpt_set_error("%s: RCCL error %d (%s)", what, (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return 1000 + (int)r;
}

int nccl_type(int dtype, ncclDataType_t* t, const char* who) {
  switch (dtype) {
    case GPT_F32: *t = ncclFloat32; return 0;
    case GPT_BF16: *t = ncclBfloat16; return 0;
    case GPT_DP_I64: *t = ncclInt64; return 0;
  }
  gpt_set_error("%s: bad dtype %d (GPT_F32, GPT_BF16 or GPT_DP_I64)", who, dtype);
  return -1;
}

#define NEED_COMM(who) GPT_REQUIRE(g_comm, "%s: no communicator (gpt_dp_init first)", who)

}  // namespace

extern "C" int gpt_dp_unique_id(void* id_out) {
  GPT_REQUIRE(id_out, "gpt_dp_unique_id: null output");
  if (int rc = load_rccl()) return rc;
  static_assert(sizeof(ncclUniqueId) == GPT_DP_ID_BYTES, "unique id size");
  ncclUniqueId id;
  if (int rc = rccl_status(g_rccl.GetUniqueId(&id), "gpt_dp_unique_id")) return rc;
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

extern "C" int gpt_dp_init(const void* id, int rank, int world) {
  GPT_REQUIRE(id && world >= 1 && rank >= 0 && rank < world, "gpt_dp_init: bad rank %d of %d (or null id)", rank, world);
  GPT_REQUIRE(!g_comm, "gpt_dp_init: a communicator exists already (gpt_dp_destroy first)");
  if (int rc = load_rccl()) return rc;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  if (int rc = rccl_status(g_rccl.CommInitRank(&g_comm, world, uid, rank), "gpt_dp_init")) {
    g_comm = nullptr;
    return rc;
  }
  g_rank = rank;
  g_world = world;
  return 0;
}

extern "C" int gpt_dp_destroy(void) {
  if (!g_comm) return 0;
  const int rc = rccl_status(g_rccl.CommDestroy(g_comm), "gpt_dp_destroy");
  g_comm = nullptr;
  g_rank = 0;
  g_world = 1;
  return rc;
}

extern "C" int gpt_dp_rank(void) { return g_rank; }
extern "C" int gpt_dp_world(void) { return g_world; }

extern "C" int gpt_dp_allreduce_grads(float* grads, int64_t n, void* stream) {
  NEED_COMM("gpt_dp_allreduce_grads");
  GPT_REQUIRE(grads && n > 0, "gpt_dp_allreduce_grads: empty buffer");
  return rccl_status(g_rccl.AllReduce(grads, grads, (size_t)n, ncclFloat32, ncclSum, g_comm, (hipStream_t)stream), "gpt_dp_allreduce_grads");
}

extern "C" int gpt_dp_reduce_scatter(const void* send, void* recv, int64_t recv_count, int dtype, void* stream) {
  NEED_COMM("gpt_dp_reduce_scatter");
  GPT_REQUIRE(send && recv && recv_count > 0, "gpt_dp_reduce_scatter: empty buffer");
  ncclDataType_t t;
  if (int rc = nccl_type(dtype, &t, "gpt_dp_reduce_scatter")) return rc;
  return rccl_status(g_rccl.ReduceScatter(send, recv, (size_t)recv_count, t, ncclSum, g_comm, (hipStream_t)stream), "gpt_dp_reduce_scatter");
}

extern "C" int gpt_dp_allgather(const void* send, void* recv, int64_t send_count, int dtype, void* stream) {
  NEED_COMM("gpt_dp_allgather");
  GPT_REQUIRE(send && recv && send_count > 0, "gpt_dp_allgather: empty buffer");
  ncclDataType_t t;
  if (int rc = nccl_type(dtype, &t, "gpt_dp_allgather")) return rc;
  return rccl_status(g_rccl.AllGather(send, recv, (size_t)send_count, t, g_comm, (hipStream_t)stream), "gpt_dp_allgather");
}

extern "C" int gpt_dp_broadcast(void* buf, int64_t count, int dtype, int root, void* stream) {
  NEED_COMM("gpt_dp_broadcast");
  GPT_REQUIRE(buf && count > 0 && root >= 0 && root < g_world, "gpt_dp_broadcast: bad arguments");
  ncclDataType_t t;
  if (int rc = nccl_type(dtype, &t, "gpt_dp_broadcast")) return rc;
  return rccl_status(g_rccl.Broadcast(buf, buf, (size_t)count, t, root, g_comm, (hipStream_t)stream), "gpt_dp_broadcast");
}

extern "C" int gpt_dp_barrier(void* stream) {
  NEED_COMM("gpt_dp_barrier");
  void* word = nullptr;      // this device's copy of the word (the array's name is a host-side stand-in, not a device address)
  const hipError_t e = hipGetSymbolAddress(&word, HIP_SYMBOL(g_barrier_word));
  GPT_REQUIRE(e == hipSuccess && word, "gpt_dp_barrier: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
  if (int rc = rccl_status(g_rccl.AllReduce(word, word, 1, ncclFloat32, ncclSum, g_comm, (hipStream_t)stream), "gpt_dp_barrier")) return rc;
  const hipError_t s = hipStreamSynchronize((hipStream_t)stream);
  GPT_REQUIRE(s == hipSuccess, "gpt_dp_barrier: %s", hipGetErrorString(s));
  return 0;
}
