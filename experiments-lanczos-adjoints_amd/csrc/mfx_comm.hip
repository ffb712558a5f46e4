// libmfx: native communicator of the row-sharded Krylov drivers -- RCCL issued by the library itself, on the caller's stream.
//
// The reference has no collective; the row-sharded drivers turn its inner products (`Q.T @ v`, arnoldi.py:87-95,
// and the adjoint's P lambda, z^T Q, :204,212) into per-rank partial sums + a sum-all-reduce, and its matvec input into an
// all-gather of the row shards (the reference's own row partition of the Gram matvec, util/gp_util.py:496-509).  Round 2 routed
// both through function pointers into Python (`libmfx -> ctypes callback -> torch.distributed`, ~280 callbacks per step); here the
// same `mfx_comm` function pointers are C functions of this file that call ncclAllReduce / ncclAllGather directly:
// no interpreter on the path, nothing between two kernels of a Krylov step but the collective itself.
//
// RCCL is bound at run time (dlopen "librccl.so.1", the SONAME PyTorch-ROCm's own copy carries too, so a process that already
// holds one does not get a second): libmfx.so has no link-time dependency on it and single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <stdio.h>
#include <string.h>

#include <mutex>

#include "mfx_internal.h"

namespace mfx {
namespace {

struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;        // optional (mfx_comm_rccl_count)
  decltype(&ncclCommUserRank) CommUserRank = nullptr;  // optional
};

char g_rccl_why[256] = "unknown";  // why rccl() is null: written once, inside the call_once below

RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
      api.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
      const char* e = dlerror();  // (null unless the failure was the most recent dl* call: read it right here)
      snprintf(g_rccl_why, sizeof(g_rccl_why), "librccl.so.1 could not be loaded: %s", e ? e : "unknown");
    }
    if (!api.lib) return;
#define MFX_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, name))
    MFX_SYM(GetUniqueId, "ncclGetUniqueId");
    MFX_SYM(CommInitRank, "ncclCommInitRank");
    MFX_SYM(CommDestroy, "ncclCommDestroy");
    MFX_SYM(AllReduce, "ncclAllReduce");
    MFX_SYM(AllGather, "ncclAllGather");
    MFX_SYM(GroupStart, "ncclGroupStart");
    MFX_SYM(GroupEnd, "ncclGroupEnd");
    MFX_SYM(GetErrorString, "ncclGetErrorString");
    MFX_SYM(CommCount, "ncclCommCount");
    MFX_SYM(CommUserRank, "ncclCommUserRank");
#undef MFX_SYM
    if (!(api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.AllGather && api.GroupStart && api.GroupEnd)) {
      snprintf(g_rccl_why, sizeof(g_rccl_why), "librccl was loaded but lacks one of the entry points used here (ncclGetUniqueId ... ncclGroupEnd)");
      dlclose(api.lib);
      api.lib = nullptr;
    }
  });
  return api.lib ? &api : nullptr;
}

struct NativeComm {
  ncclComm_t comm;
};

inline ncclDataType_t nccl_type(int dtype) { return dtype == MFX_F64 ? ncclDouble : ncclFloat; }

#define MFX_NCCL(expr)                                                                                            \
  do {                                                                                                            \
    ncclResult_t _r = (expr);                                                                                     \
    if (_r != ncclSuccess) {                                                                                      \
      set_error("%s failed: %s", #expr, rccl()->GetErrorString ? rccl()->GetErrorString(_r) : "RCCL error");       \
      return MFX_ERR_CALLBACK;                                                                                    \
    }                                                                                                             \
  } while (0)

int native_allreduce(void* ctx, void* buf, int64_t count, int dtype, void* stream) {
  MFX_NCCL(rccl()->AllReduce(buf, buf, (size_t)count, nccl_type(dtype), ncclSum, static_cast<NativeComm*>(ctx)->comm,
                             static_cast<hipStream_t>(stream)));
  return 0;
}

int native_allgather(void* ctx, const void* in, void* out, int64_t count, int dtype, void* stream) {
  MFX_NCCL(rccl()->AllGather(in, out, (size_t)count, nccl_type(dtype), static_cast<NativeComm*>(ctx)->comm,
                             static_cast<hipStream_t>(stream)));
  return 0;
}

// p vectors at once, straight from the row shards into the (p, n) operator input: vector b of every rank lands at
// full[b][rank * count ...] -- one grouped launch, no pack / unpack copy on either side
int native_allgather_rows(void* ctx, const void* local, int64_t ldlocal, void* full, int64_t ldfull, int64_t p, int64_t count,
                          int dtype, void* stream) {
  const size_t es = dtype == MFX_F64 ? 8 : 4;
  RcclApi* a = rccl();
  MFX_NCCL(a->GroupStart());
  for (int64_t b = 0; b < p; ++b) {
    ncclResult_t r = a->AllGather(static_cast<const char*>(local) + b * ldlocal * es, static_cast<char*>(full) + b * ldfull * es,
                                  (size_t)count, nccl_type(dtype), static_cast<NativeComm*>(ctx)->comm, static_cast<hipStream_t>(stream));
    if (r != ncclSuccess) {
      a->GroupEnd();
      set_error("ncclAllGather (vector %lld) failed: %s", (long long)b, a->GetErrorString ? a->GetErrorString(r) : "RCCL error");
      return MFX_ERR_CALLBACK;
    }
  }
  MFX_NCCL(a->GroupEnd());
  return 0;
}

}  // namespace
}  // namespace mfx

extern "C" {

int mfx_rccl_available(void) { return mfx::rccl() != nullptr ? 1 : 0; }

int mfx_rccl_unique_id(void* id, int64_t bytes) {
  using namespace mfx;
  MFX_REQUIRE(id && bytes >= (int64_t)sizeof(ncclUniqueId), MFX_ERR_INVALID, "unique-id buffer of %lld bytes, need %zu", (long long)bytes,
              sizeof(ncclUniqueId));
  RcclApi* a = rccl();
  MFX_REQUIRE(a, MFX_ERR_UNSUPPORTED, "%s", g_rccl_why);
  ncclUniqueId uid;
  MFX_NCCL(a->GetUniqueId(&uid));
  memcpy(id, &uid, sizeof(uid));
  return MFX_OK;
}

int mfx_comm_create_rccl(const void* id, int64_t bytes, int32_t rank, int32_t world, int64_t nloc, mfx_comm* out) {
  using namespace mfx;
  MFX_REQUIRE(id && out && bytes >= (int64_t)sizeof(ncclUniqueId), MFX_ERR_INVALID, "null argument or short unique id");
  MFX_REQUIRE(world >= 1 && rank >= 0 && rank < world && nloc >= 1, MFX_ERR_INVALID, "rank %d of %d, nloc %lld", rank, world, (long long)nloc);
  RcclApi* a = rccl();
  MFX_REQUIRE(a, MFX_ERR_UNSUPPORTED, "%s", g_rccl_why);
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  NativeComm* nc = new NativeComm();
  ncclResult_t r = a->CommInitRank(&nc->comm, world, uid, rank);  // collective over the `world` ranks; binds the CURRENT device
  if (r != ncclSuccess) {
    delete nc;
    set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, a->GetErrorString ? a->GetErrorString(r) : "RCCL error");
    return MFX_ERR_CALLBACK;
  }
  out->rank = rank;
  out->world = world;
  out->nloc = nloc;
  out->allreduce_sum = native_allreduce;
  out->allgather = native_allgather;
  out->ctx = nc;
  out->exchange = nullptr;
  out->allgather_rows = native_allgather_rows;
  return MFX_OK;
}

int mfx_comm_rccl_gather_mode(mfx_comm* comm, int mode) {
  using namespace mfx;
  MFX_REQUIRE(comm && comm->ctx && comm->allreduce_sum == native_allreduce, MFX_ERR_INVALID, "not a communicator of mfx_comm_create_rccl");
  MFX_REQUIRE(mode == MFX_GATHER_GROUPED || mode == MFX_GATHER_PACKED, MFX_ERR_INVALID, "gather mode %d", mode);
  // packed: the drivers take their generic path -- k_pack_shard, ONE ncclAllGather of the (p, nloc) shard (native_allgather), k_unshard
  comm->allgather_rows = mode == MFX_GATHER_GROUPED ? native_allgather_rows : nullptr;
  return MFX_OK;
}

int mfx_comm_rccl_count(const mfx_comm* comm, int32_t* ranks, int32_t* rank) {
  using namespace mfx;
  MFX_REQUIRE(comm && comm->ctx && comm->allreduce_sum == native_allreduce, MFX_ERR_INVALID, "not a communicator of mfx_comm_create_rccl");
  MFX_REQUIRE(ranks && rank, MFX_ERR_INVALID, "null output");
  RcclApi* a = rccl();
  MFX_REQUIRE(a && a->CommCount && a->CommUserRank, MFX_ERR_UNSUPPORTED, "librccl lacks ncclCommCount / ncclCommUserRank");
  int cnt = -1, me = -1;
  MFX_NCCL(a->CommCount(static_cast<NativeComm*>(comm->ctx)->comm, &cnt));
  MFX_NCCL(a->CommUserRank(static_cast<NativeComm*>(comm->ctx)->comm, &me));
  *ranks = cnt;
  *rank = me;
  return MFX_OK;
}

int mfx_comm_destroy_rccl(mfx_comm* comm) {
  using namespace mfx;
  if (!comm || !comm->ctx || comm->allreduce_sum != native_allreduce) return MFX_OK;  // not one of ours: nothing to release
  NativeComm* nc = static_cast<NativeComm*>(comm->ctx);
  RcclApi* a = rccl();
  if (a) a->CommDestroy(nc->comm);
  delete nc;
  comm->ctx = nullptr;
  comm->allreduce_sum = nullptr;
  comm->allgather = nullptr;
  comm->allgather_rows = nullptr;
  return MFX_OK;
}

}  // extern "C"
