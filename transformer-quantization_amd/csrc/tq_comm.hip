// Raw-RCCL exchange for sharded calibration: the statistics all-reduce of
// pass_data_for_range_estimation's loop (reference utils/utils.py:47-79 is the loop being sharded; the reference
// itself has no distributed code) issued straight on librccl from C -- no torch.distributed / c10d in the data path:
//   * one ctypes call per collective (or per whole calibrating step: tq_calibrate_minmax_rccl) instead of ~37 us of
//     c10d host bookkeeping x 161 quantizer calls per BERT-base batch;
//   * no watchdog thread polling events, so a calibrating forward INCLUDING its collectives captures as a hipGraph
//     (ncclAllReduce on the capturing stream is a plain sequence of kernel launches).
// librccl is bound at run time (dlopen + dlsym) and its handful of public types are restated below (checked against
// <rccl/rccl.h> at compile time when that header is installed): libtq_hip.so builds and loads on a box without RCCL or
// without its development headers, and the process
// uses the SAME librccl that torch already mapped when there is one (tq_comm_load is handed torch/lib/librccl.so by
// the Python side) -- never two RCCL runtimes in one process.  The communicator is created from a 128-byte
// ncclUniqueId that rank 0 generates (tq_comm_get_unique_id) and the caller ships to the other ranks by any means
// (the Python side uses the torch.distributed rendezvous store); one process per GPU, one communicator per process.
#include <dlfcn.h>
#include <string.h>

#include "tq_host.h"

// The slice of the NCCL/RCCL public ABI this file calls (rccl.h: opaque communicator, 128-byte unique id, result /
// type / reduction codes).  These values are frozen by NCCL's ABI; the static_asserts below compare them with the
// installed header whenever there is one.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define TQ_HAVE_RCCL_H 1
#endif

namespace tq {
namespace nccl_abi {
struct Comm;
using comm_t = Comm*;
struct UniqueId { char internal[128]; };
using result_t = int;        // ncclSuccess == 0
constexpr int kSuccess = 0;
constexpr int kUint8 = 1, kInt32 = 2, kFloat32 = 7, kFloat64 = 8;
constexpr int kSum = 0, kMax = 2, kMin = 3;
#ifdef TQ_HAVE_RCCL_H
static_assert(sizeof(UniqueId) == sizeof(ncclUniqueId), "ncclUniqueId");
static_assert(kSuccess == ncclSuccess && kUint8 == ncclUint8 && kInt32 == ncclInt32 && kFloat32 == ncclFloat32 &&
              kFloat64 == ncclFloat64 && kSum == ncclSum && kMax == ncclMax && kMin == ncclMin, "nccl enum codes");
static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int),
              "nccl enums are int-sized");
#endif
}  // namespace nccl_abi
}  // namespace tq

namespace tq {

struct RcclApi {
  void* handle = nullptr;
  nccl_abi::result_t (*GetUniqueId)(nccl_abi::UniqueId*) = nullptr;
  nccl_abi::result_t (*CommInitRank)(nccl_abi::comm_t*, int, nccl_abi::UniqueId, int) = nullptr;
  nccl_abi::result_t (*CommDestroy)(nccl_abi::comm_t) = nullptr;
  nccl_abi::result_t (*CommAbort)(nccl_abi::comm_t) = nullptr;
  nccl_abi::result_t (*CommCount)(const nccl_abi::comm_t, int*) = nullptr;
  nccl_abi::result_t (*CommUserRank)(const nccl_abi::comm_t, int*) = nullptr;
  nccl_abi::result_t (*AllReduce)(const void*, void*, size_t, int, int, nccl_abi::comm_t, hipStream_t) = nullptr;
  nccl_abi::result_t (*Broadcast)(const void*, void*, size_t, int, int, nccl_abi::comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(nccl_abi::result_t) = nullptr;
  nccl_abi::result_t (*GetVersion)(int*) = nullptr;
};

static RcclApi g_rccl;

static const char* rccl_err(nccl_abi::result_t r) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error"; }

template <class F>
static bool bind(void* h, const char* name, F& fn) {
  fn = reinterpret_cast<F>(dlsym(h, name));
  return fn != nullptr;
}

static int load_rccl(const char* path) {
  if (g_rccl.handle) return TQ_OK;
  void* h = nullptr;
  const char* tried = path;
  if (path && *path) {
    h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  } else {
    // an already mapped librccl (torch's) first, then the loader's search path
    static const char* const names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) { tried = n; break; }
    if (!h)
      for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) { tried = n; break; }
    if (!h && (h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL))) tried = "/opt/rocm/lib/librccl.so";
  }
  if (!h) {
    const char* why = dlerror();
    return set_error(TQ_EUNSUPPORTED, "tq_comm_load: cannot open librccl (%s): %s", tried ? tried : "librccl.so",
                     why ? why : "not found");
  }
  RcclApi a;
  a.handle = h;
  int missing = 0;
  missing += !bind(h, "ncclGetUniqueId", a.GetUniqueId);
  missing += !bind(h, "ncclCommInitRank", a.CommInitRank);
  missing += !bind(h, "ncclCommDestroy", a.CommDestroy);
  missing += !bind(h, "ncclAllReduce", a.AllReduce);
  missing += !bind(h, "ncclGetErrorString", a.GetErrorString);
  missing += !bind(h, "ncclCommCount", a.CommCount);
  missing += !bind(h, "ncclCommUserRank", a.CommUserRank);
  missing += !bind(h, "ncclBroadcast", a.Broadcast);
  const bool ok = missing == 0;
  bind(h, "ncclCommAbort", a.CommAbort);
  bind(h, "ncclGetVersion", a.GetVersion);
  if (!ok) return set_error(TQ_EUNSUPPORTED, "tq_comm_load: %s lacks the nccl* entry points", tried ? tried : "librccl");
  g_rccl = a;
  return TQ_OK;
}

static int nccl_type(int dtype, int* t) {
  switch (dtype) {
    case TQ_COMM_F32: *t = nccl_abi::kFloat32; return TQ_OK;
    case TQ_COMM_F64: *t = nccl_abi::kFloat64; return TQ_OK;
    case TQ_COMM_I32: *t = nccl_abi::kInt32; return TQ_OK;
    case TQ_COMM_U8: *t = nccl_abi::kUint8; return TQ_OK;
    default: return set_error(TQ_EINVAL, "tq_comm: unknown element type %d", dtype);
  }
}

}  // namespace tq

using namespace tq;

extern "C" size_t tq_comm_unique_id_bytes(void) { return sizeof(nccl_abi::UniqueId); }

extern "C" int tq_comm_load(const char* librccl_path) { return load_rccl(librccl_path); }

extern "C" int tq_comm_version(void) {
  int v = 0;
  if (g_rccl.handle == nullptr || g_rccl.GetVersion == nullptr || g_rccl.GetVersion(&v) != nccl_abi::kSuccess) return 0;
  return v;
}

extern "C" int tq_comm_get_unique_id(void* id_out) {
  TQ_REQUIRE(id_out, "tq_comm_get_unique_id: NULL pointer");
  if (int e = load_rccl(nullptr)) return e;
  nccl_abi::UniqueId id;
  nccl_abi::result_t r = g_rccl.GetUniqueId(&id);
  if (r != nccl_abi::kSuccess) return set_error(TQ_ELAUNCH, "ncclGetUniqueId: %s", rccl_err(r));
  memcpy(id_out, &id, sizeof(id));
  return TQ_OK;
}

extern "C" int tq_comm_init(const void* unique_id, int rank, int world, void** comm_out) {
  TQ_REQUIRE(unique_id && comm_out, "tq_comm_init: NULL pointer");
  TQ_REQUIRE(world >= 1 && rank >= 0 && rank < world, "tq_comm_init: bad rank %d / world %d", rank, world);
  if (int e = load_rccl(nullptr)) return e;
  nccl_abi::UniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  nccl_abi::comm_t comm = nullptr;
  nccl_abi::result_t r = g_rccl.CommInitRank(&comm, world, id, rank);     // uses the calling thread's current device
  if (r != nccl_abi::kSuccess) return set_error(TQ_ELAUNCH, "ncclCommInitRank(rank %d of %d): %s", rank, world, rccl_err(r));
  *comm_out = comm;
  return TQ_OK;
}

extern "C" int tq_comm_destroy(void* comm) {
  if (comm == nullptr) return TQ_OK;
  TQ_REQUIRE(g_rccl.handle, "tq_comm_destroy: librccl is not loaded");
  nccl_abi::result_t r = g_rccl.CommDestroy(static_cast<nccl_abi::comm_t>(comm));
  if (r != nccl_abi::kSuccess) return set_error(TQ_ELAUNCH, "ncclCommDestroy: %s", rccl_err(r));
  return TQ_OK;
}

extern "C" int tq_comm_abort(void* comm) {
  if (comm == nullptr) return TQ_OK;
  TQ_REQUIRE(g_rccl.handle, "tq_comm_abort: librccl is not loaded");
  // ncclCommAbort does not wait for the peers (ncclCommDestroy may, when a collective of this communicator is still
  // outstanding): the call for a communicator whose set-up was rejected by the agreement round or whose peer is gone
  nccl_abi::result_t r = g_rccl.CommAbort ? g_rccl.CommAbort(static_cast<nccl_abi::comm_t>(comm))
                                          : g_rccl.CommDestroy(static_cast<nccl_abi::comm_t>(comm));
  if (r != nccl_abi::kSuccess) return set_error(TQ_ELAUNCH, "ncclCommAbort: %s", rccl_err(r));
  return TQ_OK;
}

extern "C" int tq_comm_rank_world(void* comm, int* rank, int* world) {
  TQ_REQUIRE(comm && rank && world, "tq_comm_rank_world: NULL pointer");
  TQ_REQUIRE(g_rccl.handle, "tq_comm_rank_world: librccl is not loaded");
  nccl_abi::result_t r = g_rccl.CommUserRank(static_cast<nccl_abi::comm_t>(comm), rank);
  if (r == nccl_abi::kSuccess) r = g_rccl.CommCount(static_cast<nccl_abi::comm_t>(comm), world);
  if (r != nccl_abi::kSuccess) return set_error(TQ_ELAUNCH, "ncclCommUserRank/Count: %s", rccl_err(r));
  return TQ_OK;
}

extern "C" int tq_comm_allreduce(void* comm, void* buf, uint64_t count, int dtype, int op, tq_stream_t stream) {
  TQ_REQUIRE(comm && (buf || count == 0), "tq_comm_allreduce: NULL pointer");
  TQ_REQUIRE(g_rccl.handle, "tq_comm_allreduce: librccl is not loaded");
  TQ_REQUIRE(op == TQ_COMM_MAX || op == TQ_COMM_SUM || op == TQ_COMM_MIN, "tq_comm_allreduce: unknown op %d", op);
  if (count == 0) return TQ_OK;
  int t;
  if (int e = nccl_type(dtype, &t)) return e;
  const int o = op == TQ_COMM_MAX ? nccl_abi::kMax : (op == TQ_COMM_MIN ? nccl_abi::kMin : nccl_abi::kSum);
  nccl_abi::result_t r = g_rccl.AllReduce(buf, buf, count, t, o, static_cast<nccl_abi::comm_t>(comm), static_cast<hipStream_t>(stream));
  if (r != nccl_abi::kSuccess) return set_error(TQ_ELAUNCH, "ncclAllReduce(%llu x type %d): %s", (unsigned long long)count, dtype, rccl_err(r));
  return TQ_OK;
}

extern "C" int tq_comm_broadcast(void* comm, void* buf, uint64_t count, int dtype, int root, tq_stream_t stream) {
  TQ_REQUIRE(comm && (buf || count == 0), "tq_comm_broadcast: NULL pointer");
  TQ_REQUIRE(g_rccl.handle, "tq_comm_broadcast: librccl is not loaded");
  if (count == 0) return TQ_OK;
  int t;
  if (int e = nccl_type(dtype, &t)) return e;
  nccl_abi::result_t r = g_rccl.Broadcast(buf, buf, count, t, root, static_cast<nccl_abi::comm_t>(comm), static_cast<hipStream_t>(stream));
  if (r != nccl_abi::kSuccess) return set_error(TQ_ELAUNCH, "ncclBroadcast: %s", rccl_err(r));
  return TQ_OK;
}

// The whole sharded calibrating step of one quantizer as ONE C call: local [-min | max] statistics ->
// ncclAllReduce(MAX) in place on the caller's stream -> estimator update + range->parameters (+ quantize).
// `stats` scratch: fp32 [2 * n_params] at the start of `workspace` (256-byte aligned), the rest is the statistics
// kernels' (tq_calibrate_workspace_bytes covers both).  hipGraph-capturable; no host synchronisation.
extern "C" int tq_calibrate_minmax_rccl(const void* x, uint64_t n, int dtype, uint64_t n_params, uint64_t inner, int mode,
                                        const float* prev_min, const float* prev_max, float* cur_min, float* cur_max,
                                        double momentum, uint64_t n_groups, const int64_t* order, int n_bits, int symmetric,
                                        float eps, int log_domain, float* delta, float* zero_float, uint8_t* signed_flag,
                                        void* y, void* workspace, size_t workspace_bytes, uint32_t* counter, void* comm,
                                        tq_stream_t stream) {
  TQ_REQUIRE(comm, "tq_calibrate_minmax_rccl: NULL communicator");
  const size_t stats_bytes = (2 * n_params * sizeof(float) + 255) / 256 * 256;
  TQ_REQUIRE(workspace && workspace_bytes >= stats_bytes, "tq_calibrate_minmax_rccl: workspace too small");
  float* stats = static_cast<float*>(workspace);
  char* rest = static_cast<char*>(workspace) + stats_bytes;
  int prev_in_stats = 0;
  if (int e = calibrate_stats_for_exchange(x, n, dtype, n_params, inner, stats, rest, workspace_bytes - stats_bytes, counter,
                                           prev_min, prev_max, &prev_in_stats, stream)) return e;
  if (int e = tq_comm_allreduce(comm, stats, 2 * n_params, TQ_COMM_F32, TQ_COMM_MAX, stream)) return e;
  return calibrate_apply_after_exchange(stats, x, n, dtype, n_params, inner, mode, prev_min, prev_max, cur_min, cur_max, momentum,
                                        n_groups, order, n_bits, symmetric, eps, log_domain, delta, zero_float, signed_flag, y,
                                        prev_in_stats, stream);
}
