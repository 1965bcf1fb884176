// Data-parallel gradient exchange over RCCL: one communicator per process (= per GPU), in-place
// sum all-reduce of contiguous slices ("buckets") of the flat gradient arenas on a caller-chosen HIP stream.
// The reference is single-GPU (README.md:91), so there is no reference interface to mirror: the entry points are the
// three calls SURVEY 8(b) derives (init / allreduce_bucket / finalize) plus the id hand-off and a broadcast for
// the initial replicas.  RCCL is bound at run time with dlopen so that libsradsgan_hip.so has no link-time
// dependency on it (single-GPU users never load it) and so that the process ends up with ONE copy of librccl: the one
// PyTorch already mapped, when there is one.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include "common.h"

namespace srhip {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;
static ncclComm_t g_comm = nullptr;
static int g_rank = 0, g_world = 0;

static int load_rccl() {
  if (g_rccl.handle) return SRHIP_OK;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);      // the copy already in the process (PyTorch's)
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    set_error("dp: cannot load librccl.so.1: %s", dlerror());
    return SRHIP_ERR_LAUNCH;
  }
#define SRHIP_SYM(field, name)                                        \
  do {                                                                \
    *reinterpret_cast<void**>(&g_rccl.field) = dlsym(h, name);        \
    if (!g_rccl.field) {                                              \
      set_error("dp: librccl has no symbol %s", name);                \
      return SRHIP_ERR_LAUNCH;                                        \
    }                                                                 \
  } while (0)
  SRHIP_SYM(GetUniqueId, "ncclGetUniqueId");
  SRHIP_SYM(CommInitRank, "ncclCommInitRank");
  SRHIP_SYM(CommDestroy, "ncclCommDestroy");
  SRHIP_SYM(AllReduce, "ncclAllReduce");
  SRHIP_SYM(Broadcast, "ncclBroadcast");
  SRHIP_SYM(GroupStart, "ncclGroupStart");
  SRHIP_SYM(GroupEnd, "ncclGroupEnd");
  SRHIP_SYM(GetErrorString, "ncclGetErrorString");
#undef SRHIP_SYM
  g_rccl.handle = h;
  return SRHIP_OK;
}

static int rccl_check(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return SRHIP_OK;
  set_error("%s: RCCL error %d (%s)", what, (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return SRHIP_ERR_LAUNCH;
}

}  // namespace srhip

using namespace srhip;

extern "C" {

int srhip_dp_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

int srhip_dp_unique_id(void* id_out) {
  SRHIP_REQUIRE(id_out != nullptr, "dp_unique_id: null output");
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  rc = rccl_check(g_rccl.GetUniqueId(&id), "dp_unique_id");
  if (rc) return rc;
  memcpy(id_out, &id, sizeof(id));
  return SRHIP_OK;
}

int srhip_dp_init(const void* id_in, int rank, int world) {
  SRHIP_REQUIRE(id_in != nullptr && world >= 1 && rank >= 0 && rank < world, "dp_init: bad rank %d / world %d", rank, world);
  SRHIP_REQUIRE(g_comm == nullptr, "dp_init: communicator already initialised (call srhip_dp_finalize first)");
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  memcpy(&id, id_in, sizeof(id));
  rc = rccl_check(g_rccl.CommInitRank(&g_comm, world, id, rank), "dp_init");
  if (rc) {
    g_comm = nullptr;
    return rc;
  }
  g_rank = rank;
  g_world = world;
  return SRHIP_OK;
}

int srhip_dp_world(void) { return g_comm ? g_world : 0; }
int srhip_dp_rank(void) { return g_comm ? g_rank : -1; }

int srhip_dp_allreduce_bucket(float* buf, size_t count, void* stream) {
  SRHIP_REQUIRE(g_comm != nullptr, "dp_allreduce_bucket: srhip_dp_init has not been called");
  if (count == 0) return SRHIP_OK;
  SRHIP_REQUIRE(buf != nullptr, "dp_allreduce_bucket: null buffer");
  return rccl_check(g_rccl.AllReduce(buf, buf, count, ncclFloat32, ncclSum, g_comm, as_stream(stream)), "dp_allreduce_bucket");
}

int srhip_dp_broadcast(float* buf, size_t count, int root, void* stream) {
  SRHIP_REQUIRE(g_comm != nullptr, "dp_broadcast: srhip_dp_init has not been called");
  if (count == 0) return SRHIP_OK;
  SRHIP_REQUIRE(buf != nullptr && root >= 0 && root < g_world, "dp_broadcast: bad arguments");
  return rccl_check(g_rccl.Broadcast(buf, buf, count, ncclFloat32, root, g_comm, as_stream(stream)), "dp_broadcast");
}

int srhip_dp_finalize(void) {
  if (!g_comm) return SRHIP_OK;
  ncclResult_t r = g_rccl.CommDestroy(g_comm);
  g_comm = nullptr;
  g_world = 0;
  return rccl_check(r, "dp_finalize");
}

}  // extern "C"
