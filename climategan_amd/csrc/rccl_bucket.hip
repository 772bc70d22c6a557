// Gradient-bucket all-reduce straight on RCCL (the collective of the data-parallel step) behind the
// C ABI: the host mirror's default path goes through torch.distributed (backend "nccl" = RCCL); these entry points are the
// same collective without torch in the call, for a host that owns its communicator -- and the reducer's
// opt-in direct path (CGAN_DDP_DIRECT_RCCL=1, climategan_amd/parallel.py).
//
// librccl is NOT a link-time dependency: the library must load on a box without RCCL (CPU build check, single-GPU use).
// cgan_rccl_load(path) dlopens it (the copy torch ships, so that one RCCL version serves the process) and resolves the five
// symbols used; every other entry point fails with CGAN_ERR_UNSUPPORTED until it has succeeded.
#include <dlfcn.h>
#include <string.h>

#include "cgan_common.h"

namespace {

struct UniqueId {
  char internal[128];     // NCCL_UNIQUE_ID_BYTES (rccl.h:40-43)
};
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);

void* g_handle = nullptr;
GetUniqueIdFn p_get_unique_id = nullptr;
CommInitRankFn p_comm_init_rank = nullptr;
CommDestroyFn p_comm_destroy = nullptr;
AllReduceFn p_all_reduce = nullptr;
GetErrorStringFn p_get_error_string = nullptr;

constexpr int NCCL_SUM = 0, NCCL_FLOAT32 = 7, NCCL_BFLOAT16 = 9, NCCL_FLOAT16 = 6;   // rccl.h:448-468

int fail(const char* what, int rc) {
  cgan_set_error("%s: RCCL error %d (%s)", what, rc, p_get_error_string ? p_get_error_string(rc) : "?");
  return CGAN_ERR_HIP;
}

}  // namespace

extern "C" int cgan_rccl_load(const char* path) {
  if (g_handle) return CGAN_OK;
  CGAN_REQUIRE(path != nullptr, "rccl_load: null path");
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    cgan_set_error("rccl_load: dlopen(%s) failed: %s", path, dlerror());
    return CGAN_ERR_UNSUPPORTED;
  }
  p_get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
  p_comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
  p_comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
  p_all_reduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
  p_get_error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
  if (!p_get_unique_id || !p_comm_init_rank || !p_comm_destroy || !p_all_reduce) {
    cgan_set_error("rccl_load: %s does not export the RCCL entry points", path);
    dlclose(h);
    return CGAN_ERR_UNSUPPORTED;
  }
  g_handle = h;
  return CGAN_OK;
}

extern "C" int cgan_rccl_loaded(void) { return g_handle != nullptr; }

extern "C" int cgan_comm_unique_id(void* id128) {
  CGAN_REQUIRE(g_handle != nullptr, "comm_unique_id: cgan_rccl_load has not succeeded");
  CGAN_REQUIRE(id128 != nullptr, "comm_unique_id: null pointer");
  int rc = p_get_unique_id((UniqueId*)id128);
  return rc == 0 ? CGAN_OK : fail("comm_unique_id", rc);
}

extern "C" int cgan_comm_init_rank(void** comm, int32_t nranks, const void* id128, int32_t rank) {
  CGAN_REQUIRE(g_handle != nullptr, "comm_init_rank: cgan_rccl_load has not succeeded");
  CGAN_REQUIRE(comm && id128, "comm_init_rank: null pointer");
  CGAN_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "comm_init_rank: rank %d of %d", rank, nranks);
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  int rc = p_comm_init_rank((Comm*)comm, nranks, id, rank);
  return rc == 0 ? CGAN_OK : fail("comm_init_rank", rc);
}

extern "C" int cgan_comm_destroy(void* comm) {
  CGAN_REQUIRE(g_handle != nullptr, "comm_destroy: cgan_rccl_load has not succeeded");
  if (!comm) return CGAN_OK;
  int rc = p_comm_destroy((Comm)comm);
  return rc == 0 ? CGAN_OK : fail("comm_destroy", rc);
}

// In-place SUM over the communicator's ranks of one flat gradient bucket (fp32, or a 16-bit wire format), enqueued on
// `stream`; the caller scales by 1 / world afterwards (GradBucketReducer.finish).  reference: the gradient averaging the
// single-process trainer gets for free on its concatenated batch (trainer.py:674-683).
extern "C" int cgan_allreduce_bucket(void* buf, int64_t count, int32_t dtype, void* comm, void* stream) {
  CGAN_REQUIRE(g_handle != nullptr, "allreduce_bucket: cgan_rccl_load has not succeeded");
  CGAN_REQUIRE(buf && comm, "allreduce_bucket: null pointer");
  CGAN_REQUIRE(count > 0, "allreduce_bucket: empty bucket");
  int dt;
  if (dtype == CGAN_F32) dt = NCCL_FLOAT32;
  else if (dtype == CGAN_BF16) dt = NCCL_BFLOAT16;
  else if (dtype == CGAN_F16) dt = NCCL_FLOAT16;
  else {
    cgan_set_error("allreduce_bucket: bad dtype %d", dtype);
    return CGAN_ERR_UNSUPPORTED;
  }
  int rc = p_all_reduce(buf, buf, (size_t)count, dt, NCCL_SUM, (Comm)comm, (hipStream_t)stream);
  return rc == 0 ? CGAN_OK : fail("allreduce_bucket", rc);
}
