// Latency-optimised MAX all-reduce of <= 8 KB over peer-to-peer mapped memory ("mailbox"), SURVEY.md 8(e):
// sharded calibration exchanges one tiny [-min | max] vector per quantizer call (8 B per-tensor, 6 KB per-embedding),
// 161 times per BERT-base batch, inline -- the collective's latency, not its bandwidth, is what counts.  Through
// torch.distributed one such ncclAllReduce costs ~37 us of host time (c10d bookkeeping + an extra kernel); this is
// ONE small kernel between the statistics kernel and the update kernel and no host work.
//
// Protocol (one process per GPU; every rank owns a mailbox in its own HBM, mapped by all peers through
// hipIpc handles over xGMI): "write local, read remote".
//   post : rank r copies its vector into slot (seq & 1) of ITS OWN mailbox, makes it visible at system scope
//          (__threadfence_system: L2 write-back), then stores seq into the slot's flag.
//   wait : for every peer p, spin on p's flag (system-scope loads of remote memory) until it shows seq, then fold
//          p's payload with fmaxf.  Everybody computes the same result; no second round.
// seq lives in the mailbox itself and is incremented by the kernel, so a captured hipGraph replays correctly.
// Two slots suffice: a peer can be at most one call ahead (it cannot finish call k+1 before this rank posted k+1,
// which happens after this rank finished reading call k).  The spin is bounded: on timeout the result is NaN and
// a status word is raised -- the kernel can never hang the device (the Python side self-tests the path against
// RCCL when it is enabled and falls back to RCCL if anything is off).
//
// The mailbox memory is the ONE allocation this library makes itself (fine-grained device memory + IPC handle):
// it must outlive every launch and be mapped into other processes, which a caller-owned torch tensor cannot promise.
#include <string.h>

#include "tq_device.h"
#include "tq_host.h"

namespace tq {

constexpr uint32_t kMailPayloadFloats = 2048;                           // 8 KB
constexpr size_t kMailHeader = 256;                                     // seq counter (+ padding)
constexpr size_t kMailSlot = 256 + kMailPayloadFloats * sizeof(float);  // flag (+ padding) | payload
constexpr size_t kMailBytes = 32768;
static_assert(kMailHeader + 2 * kMailSlot <= kMailBytes, "mailbox layout");
// default spin budget: one poll is a system-scope load of remote memory + s_sleep, ~1 us -> ~10 minutes, the order of
// c10d's collective timeout (ordinary rank skew -- a data-loader stall, first-batch work on rank 0 -- must never time out;
// RCCL would simply wait).  The Python side reads `status` when calibration ends and raises.
constexpr uint32_t kDefaultSpin = 600000000u;

__device__ __forceinline__ uint32_t* mail_flag(void* base, uint32_t parity) {
  return reinterpret_cast<uint32_t*>(static_cast<char*>(base) + kMailHeader + parity * kMailSlot);
}
__device__ __forceinline__ float* mail_payload(void* base, uint32_t parity) {
  return reinterpret_cast<float*>(static_cast<char*>(base) + kMailHeader + parity * kMailSlot + 256);
}

__global__ __launch_bounds__(kBlock) void mailbox_allreduce_max_k(float* __restrict__ stats, uint32_t n, void* my_base,
                                                                  void* const* __restrict__ peers, uint32_t world,
                                                                  uint32_t rank, uint32_t* __restrict__ status,
                                                                  uint32_t spin_budget) {
  constexpr int PER = kMailPayloadFloats / kBlock;       // 8 values per lane
  __shared__ uint32_t s_seq, s_ok;
  if (threadIdx.x == 0) {
    uint32_t* counter = static_cast<uint32_t*>(my_base);
    s_seq = *counter + 1;
    *counter = s_seq;
    s_ok = 1;
  }
  __syncthreads();
  const uint32_t seq = s_seq, parity = seq & 1u;
  float acc[PER];
  float* mine = mail_payload(my_base, parity);
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const uint32_t i = threadIdx.x + j * kBlock;
    acc[j] = i < n ? stats[i] : 0.0f;
    if (i < n) mine[i] = acc[j];
  }
  __threadfence_system();                                 // payload visible beyond this device before the flag
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(mail_flag(my_base, parity), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);

  for (uint32_t p = 0; p < world; ++p) {
    if (p == rank) continue;
    void* pb = peers[p];
    if (threadIdx.x == 0) {
      uint32_t budget = spin_budget;
      const uint32_t* f = mail_flag(pb, parity);
      while ((int32_t)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
        if (--budget == 0) { s_ok = 0; break; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    if (!s_ok) break;
    __threadfence_system();
    const float* theirs = mail_payload(pb, parity);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const uint32_t i = threadIdx.x + j * kBlock;
      if (i < n) {
        const float v = __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // never from a stale L2 line
        acc[j] = max_nanprop(acc[j], v);
      }
    }
    __syncthreads();
  }
  const bool ok = s_ok != 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const uint32_t i = threadIdx.x + j * kBlock;
    if (i < n) stats[i] = ok ? acc[j] : __builtin_nanf("");
  }
  if (!ok && threadIdx.x == 0) atomicOr(status, 1u);
}

}  // namespace tq

using namespace tq;

extern "C" size_t tq_mailbox_bytes(void) { return kMailBytes; }
extern "C" size_t tq_mailbox_max_floats(void) { return kMailPayloadFloats; }
extern "C" size_t tq_mailbox_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }

extern "C" int tq_mailbox_alloc(void** base, void* ipc_handle_out) {
  TQ_REQUIRE(base && ipc_handle_out, "tq_mailbox_alloc: NULL pointer");
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, kMailBytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipMalloc(&p, kMailBytes);
  }
  if (e != hipSuccess) return set_error(TQ_ELAUNCH, "tq_mailbox_alloc: %s", hipGetErrorString(e));
  e = hipMemset(p, 0, kMailBytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) {
    (void)hipFree(p);
    return set_error(TQ_ELAUNCH, "tq_mailbox_alloc: %s (multi-process GPU work needs HSA_ENABLE_IPC_MODE_LEGACY=0)",
                     hipGetErrorString(e));
  }
  memcpy(ipc_handle_out, &h, sizeof(h));
  *base = p;
  return TQ_OK;
}

extern "C" int tq_mailbox_open(const void* ipc_handle, void** peer_base) {
  TQ_REQUIRE(ipc_handle && peer_base, "tq_mailbox_open: NULL pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, ipc_handle, sizeof(h));
  void* p = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return set_error(TQ_ELAUNCH, "tq_mailbox_open: %s", hipGetErrorString(e));
  *peer_base = p;
  return TQ_OK;
}

extern "C" int tq_mailbox_close(void* peer_base) {
  if (peer_base == nullptr) return TQ_OK;
  hipError_t e = hipIpcCloseMemHandle(peer_base);
  if (e != hipSuccess) return set_error(TQ_ELAUNCH, "tq_mailbox_close: %s", hipGetErrorString(e));
  return TQ_OK;
}

extern "C" int tq_mailbox_free(void* base) {
  if (base == nullptr) return TQ_OK;
  hipError_t e = hipFree(base);
  if (e != hipSuccess) return set_error(TQ_ELAUNCH, "tq_mailbox_free: %s", hipGetErrorString(e));
  return TQ_OK;
}

extern "C" int tq_mailbox_allreduce_max(float* stats, uint64_t n, void* my_base, void* const* peer_bases, uint32_t world,
                                        uint32_t rank, uint32_t* status, uint32_t spin_budget, tq_stream_t stream) {
  TQ_REQUIRE(stats && my_base && peer_bases && status, "tq_mailbox_allreduce_max: NULL pointer");
  TQ_REQUIRE(n >= 1 && n <= kMailPayloadFloats, "tq_mailbox_allreduce_max: n=%llu outside 1..%u", (unsigned long long)n,
             kMailPayloadFloats);
  TQ_REQUIRE(world >= 1 && rank < world, "tq_mailbox_allreduce_max: bad rank %u / world %u", rank, world);
  hipLaunchKernelGGL(mailbox_allreduce_max_k, dim3(1), dim3(kBlock), 0, static_cast<hipStream_t>(stream), stats, (uint32_t)n,
                     my_base, peer_bases, world, rank, status, spin_budget ? spin_budget : kDefaultSpin);
  return check_launch("mailbox_allreduce_max_k");
}

// The whole sharded calibrating step as ONE C call: local statistics -> mailbox MAX all-reduce -> estimator update +
// parameters (+ quantize): 3-4 launches, no host work in between (the Python side then costs what the single-GPU
// fused step costs).  `stats` scratch: fp32 [2 * n_params] at the start of `workspace`, 256-byte aligned; the rest of
// the workspace is the statistics kernels' (tq_calibrate_workspace_bytes covers both).
extern "C" int tq_calibrate_minmax_mailbox(const void* x, uint64_t n, int dtype, uint64_t n_params, uint64_t inner, int mode,
                                           const float* prev_min, const float* prev_max, float* cur_min, float* cur_max,
                                           double momentum, uint64_t n_groups, const int64_t* order, int n_bits,
                                           int symmetric, float eps, int log_domain, float* delta, float* zero_float,
                                           uint8_t* signed_flag, void* y, void* workspace, size_t workspace_bytes,
                                           uint32_t* counter, void* my_base, void* const* peer_bases, uint32_t world,
                                           uint32_t rank, uint32_t* status, uint32_t spin_budget, tq_stream_t stream) {
  TQ_REQUIRE(2 * n_params <= kMailPayloadFloats, "tq_calibrate_minmax_mailbox: %llu ranges do not fit the mailbox",
             (unsigned long long)n_params);
  const size_t stats_bytes = (2 * n_params * sizeof(float) + 255) / 256 * 256;
  TQ_REQUIRE(workspace && workspace_bytes >= stats_bytes, "tq_calibrate_minmax_mailbox: workspace too small");
  float* stats = static_cast<float*>(workspace);
  char* rest = static_cast<char*>(workspace) + stats_bytes;
  int prev_in_stats = 0;
  if (int e = calibrate_stats_for_exchange(x, n, dtype, n_params, inner, stats, rest, workspace_bytes - stats_bytes, counter,
                                           prev_min, prev_max, &prev_in_stats, stream)) return e;
  if (int e = tq_mailbox_allreduce_max(stats, 2 * n_params, my_base, peer_bases, world, rank, status, spin_budget, stream)) return e;
  return calibrate_apply_after_exchange(stats, x, n, dtype, n_params, inner, mode, prev_min, prev_max, cur_min, cur_max, momentum,
                                        n_groups, order, n_bits, symmetric, eps, log_domain, delta, zero_float, signed_flag, y,
                                        prev_in_stats, stream);
}
