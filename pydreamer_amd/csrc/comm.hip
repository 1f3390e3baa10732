// The data-parallel exchange step as a NATIVE entry point (`dm_allreduce_grads(buf, n, ncclComm_t, hipStream_t)`;
// 8(e): one SUM all-reduce per optimizer group over its flat fp32 gradient buffer, RCCL over xGMI).  The reference has no
// counterpart (pydreamer is single-process); the Python path (pydreamer_amd/dist.py) issues the same collective
// through torch.distributed.  This file binds RCCL with dlopen at first use - the library has NO link-time dependency on
// librccl, so a single-GPU deployment never loads it - and gives every optimizer group its own communicator, so a group's
// all-reduce is ordered by the STREAM it is enqueued on (right behind the backward pass that filled the buffer) and by nothing
// else: no host-side drain of the launcher thread, no cross-group issue order to keep equal on all ranks.
//
// Status: executed on hardware with ONE rank only (a 1-rank communicator on the 1-GPU boxes this build has had:
// tests/test_gpu_dist.py::test_native_rccl_entry_points_one_rank); no multi-GPU node has been available in six rounds.
#include "common.h"
#include <dlfcn.h>
#include <mutex>
#include <string.h>

namespace {
struct DmNcclId { char internal[128]; };      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128), passed BY VALUE to ncclCommInitRank
typedef int (*fn_get_version)(int*);
typedef int (*fn_get_unique_id)(DmNcclId*);
typedef int (*fn_comm_init_rank)(void**, int, DmNcclId, int);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_error_string)(int);

struct Rccl {
  void* handle = nullptr;
  fn_get_version get_version = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_error_string error_string = nullptr;
  bool tried = false;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

// The process may already hold RCCL (torch links its own copy, soname librccl.so.1): take THAT instance first, so both paths
// share one library state; only then load one by name (DM_RCCL_LIB overrides the search).
bool rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.tried) return g_rccl.handle != nullptr;
  g_rccl.tried = true;
  const char* env = getenv("DM_RCCL_LIB");
  const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (int pass = 0; pass < 2 && !h; ++pass)
    for (const char* n : names) {
      if (!n || !*n) continue;
      h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (h) break;
    }
  if (!h) return false;
  g_rccl.get_version = (fn_get_version)dlsym(h, "ncclGetVersion");
  g_rccl.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
  g_rccl.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
  g_rccl.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
  g_rccl.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
  if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_reduce) return false;
  g_rccl.handle = h;
  return true;
}
int rccl_fail(const char* what, int rc) {
  return dm_fail(DM_E_HIP, "%s: RCCL error %d (%s)", what, rc, g_rccl.error_string ? g_rccl.error_string(rc) : "?");
}
}  // namespace

// 1 when librccl could be bound (already in the process, or loadable by name), else 0.  Never fails.
extern "C" int dm_rccl_available(void) { return rccl_load() ? 1 : 0; }

// ncclGetVersion's code (e.g. 22105), 0 when RCCL is not available.
extern "C" int dm_rccl_version(void) {
  if (!rccl_load() || !g_rccl.get_version) return 0;
  int v = 0;
  return g_rccl.get_version(&v) == 0 ? v : 0;
}

// Rank 0 makes the 128-byte id of a new communicator; the host side carries it to the other ranks (dist.py: through the
// torch.distributed store the job already has).
extern "C" int dm_rccl_unique_id(void* id128) {
  DM_REQUIRE(id128, DM_E_NULL, "rccl_unique_id: null argument");
  if (!rccl_load()) return dm_fail(DM_E_DEVICE, "rccl_unique_id: librccl is not loadable (DM_RCCL_LIB names it)");
  DmNcclId id;
  const int rc = g_rccl.get_unique_id(&id);
  if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(id128, id.internal, sizeof id.internal);
  return DM_OK;
}

// A communicator of `nranks` processes bound to the CURRENT device (one process per GPU); collective over all ranks.
extern "C" int dm_rccl_comm_init(void** comm, int nranks, const void* id128, int rank) {
  DM_REQUIRE(comm && id128, DM_E_NULL, "rccl_comm_init: null argument");
  DM_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, DM_E_SHAPE, "rccl_comm_init: rank %d of %d", rank, nranks);
  if (!rccl_load()) return dm_fail(DM_E_DEVICE, "rccl_comm_init: librccl is not loadable (DM_RCCL_LIB names it)");
  DmNcclId id;
  memcpy(id.internal, id128, sizeof id.internal);
  void* c = nullptr;
  const int rc = g_rccl.comm_init_rank(&c, nranks, id, rank);
  if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
  *comm = c;
  return DM_OK;
}

extern "C" int dm_rccl_comm_destroy(void* comm) {
  if (!comm) return DM_OK;
  if (!rccl_load()) return dm_fail(DM_E_DEVICE, "rccl_comm_destroy: librccl is not loadable");
  const int rc = g_rccl.comm_destroy(comm);
  return rc == 0 ? DM_OK : rccl_fail("ncclCommDestroy", rc);
}

// buf[0..n) <- sum over ranks, in place, fp32, enqueued on `stream` (never synchronises): ordered behind whatever produced
// buf on that stream and in front of whatever the stream runs next (clip + AdamW of the group).  
extern "C" int dm_allreduce_grads(void* buf, size_t n, void* comm, void* stream) {
  DM_REQUIRE(buf && comm, DM_E_NULL, "allreduce_grads: null argument");
  if (n == 0) return DM_OK;
  if (!rccl_load()) return dm_fail(DM_E_DEVICE, "allreduce_grads: librccl is not loadable");
  const int rc = g_rccl.all_reduce(buf, buf, n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm, (hipStream_t)stream);
  return rc == 0 ? DM_OK : rccl_fail("ncclAllReduce", rc);
}
