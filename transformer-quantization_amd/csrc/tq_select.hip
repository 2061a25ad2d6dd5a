// Order statistics on the device: the percentile branch of the min/max estimators.
//
// CurrentMinMaxEstimator with `percentile` (reference quantization/range_estimators.py:121-140) moves the whole
// tensor to the HOST and calls np.percentile -- a full sort of e.g. 786 432 activations per call.  np.percentile
// (method 'linear') needs exactly TWO order statistics per requested percentile: the elements at floor and ceil of the
// virtual index (n - 1) p / 100.  This file selects them without sorting: MSB-first radix select over a monotone
// 32-bit key of the fp32 value, 8 bits per pass, up to TQ_OSTAT_MAX_RANKS ranks at once (p and 100 - p need 4).
//   key(f) = bits ^ 0x80000000 for f >= +0, ~bits for negative f, 0xffffffff for NaN: ascending keys == ascending
//   values, NaN last (like torch.sort / np.sort); the value is recovered exactly from the final key.
// Two shapes of work:
//   * few long rows (per-tensor activation statistics, one row of B*T*d elements): every pass is a grid-wide
//     histogram launch (LDS histograms per block, one atomicAdd per non-empty bucket into the row's global histogram)
//     followed by a one-block-per-row selection launch that narrows each rank's (prefix, remaining rank);
//   * many short rows (per-channel weight statistics, [3072, 768]): one block per row runs the four passes itself on
//     LDS histograms; the row is re-read from L2.
// HBM traffic: 4 reads of x (16 B/elem fp32) against sort's ~10 passes of keys + values; more to the point, no
// 786 432-element device sort and no host round trip of the tensor.
#include <algorithm>

#include "tq_device.h"
#include "tq_host.h"

namespace tq {

constexpr int kMaxRanks = 4;          // == TQ_OSTAT_MAX_RANKS

__device__ __forceinline__ uint32_t ostat_key(float f) {
  const uint32_t b = __builtin_bit_cast(uint32_t, f);
  if (f != f) return 0xffffffffu;
  return (b & 0x80000000u) ? ~b : (b ^ 0x80000000u);
}
__device__ __forceinline__ float ostat_value(uint32_t k) {
  const uint32_t b = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
  return __builtin_bit_cast(float, b);       // key 0xffffffff -> 0x7fffffff, a NaN
}

struct OstatRanks {                     // the requested ranks travel by value (kernel argument): no upload, capturable
  uint64_t r[kMaxRanks];
};

struct OstatState {                     // per (row, rank)
  uint32_t prefix;                      // key bits fixed so far (high bytes)
  uint32_t rank;                        // rank among the elements that share the prefix
};

// bucket of `rank` in a 256-bin histogram held one bin per thread (blockDim.x == 256): exclusive scan in LDS
__device__ __forceinline__ void ostat_pick(uint32_t count, uint32_t rank, uint32_t* scan /* LDS [256] */, uint32_t* out_bucket,
                                           uint32_t* out_rank /* LDS scalars */) {
  const int t = threadIdx.x;
  scan[t] = count;
  __syncthreads();
#pragma unroll
  for (int o = 1; o < 256; o <<= 1) {
    const uint32_t v = t >= o ? scan[t - o] : 0u;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  const uint32_t incl = scan[t], excl = incl - count;
  if (rank >= excl && rank < incl) { *out_bucket = (uint32_t)t; *out_rank = rank - excl; }
  __syncthreads();
}

// ---- few long rows -------------------------------------------------------------------------------------------------------
// grid (blocks_per_row, rows).  hist: [rows][m][256] for THIS pass (zeroed by the caller); state: [rows][m].
template <int DT>
__global__ __launch_bounds__(256) void ostat_hist_k(const void* __restrict__ x, uint64_t n, uint32_t m, int shift,
                                                    const OstatState* __restrict__ state, uint32_t* __restrict__ hist) {
  typedef typename Store<DT>::elem_t E;
  __shared__ uint32_t h[kMaxRanks][256];
  const uint32_t row = blockIdx.y;
  for (uint32_t j = 0; j < m; ++j) h[j][threadIdx.x] = 0;
  uint32_t prefix[kMaxRanks];
  for (uint32_t j = 0; j < kMaxRanks; ++j) prefix[j] = j < m ? state[(uint64_t)row * m + j].prefix : 0u;
  const uint32_t mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
  __syncthreads();
  const E* xr = static_cast<const E*>(x) + (uint64_t)row * n;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const uint32_t k = ostat_key(Store<DT>::load1(xr + i));
    const uint32_t b = (k >> shift) & 255u;
    // ranks that still share their prefix (neighbouring order statistics usually do until the last pass) count once each:
    // the histograms are per rank, so no de-duplication is needed for correctness
#pragma unroll
    for (uint32_t j = 0; j < kMaxRanks; ++j)
      if (j < m && (k & mask) == prefix[j]) atomicAdd(&h[j][b], 1u);
  }
  __syncthreads();
  for (uint32_t j = 0; j < m; ++j) {
    const uint32_t c = h[j][threadIdx.x];
    if (c) atomicAdd(hist + ((uint64_t)row * m + j) * 256 + threadIdx.x, c);
  }
}

// one block per row: narrow every rank by the histogram of this pass; after the last pass write the values
__global__ __launch_bounds__(256) void ostat_select_k(const uint32_t* __restrict__ hist, OstatState* __restrict__ state, uint32_t m,
                                                      int shift, float* __restrict__ out) {
  __shared__ uint32_t scan[256];
  __shared__ uint32_t bucket, rest;
  const uint32_t row = blockIdx.x;
  for (uint32_t j = 0; j < m; ++j) {
    const OstatState s = state[(uint64_t)row * m + j];
    ostat_pick(hist[((uint64_t)row * m + j) * 256 + threadIdx.x], s.rank, scan, &bucket, &rest);
    if (threadIdx.x == 0) {
      const uint32_t prefix = s.prefix | (bucket << shift);
      state[(uint64_t)row * m + j] = OstatState{prefix, rest};
      if (shift == 0) out[(uint64_t)row * m + j] = ostat_value(prefix);
    }
    __syncthreads();
  }
}

__global__ void ostat_init_k(OstatState* __restrict__ state, OstatRanks ranks, uint64_t rows, uint32_t m) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows * m) state[i] = OstatState{0u, (uint32_t)ranks.r[i % m]};
}

// ---- many short rows: one block per row, four passes on LDS histograms -------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void ostat_rows_k(const void* __restrict__ x, uint64_t n, uint32_t m, OstatRanks ranks,
                                                    float* __restrict__ out) {
  typedef typename Store<DT>::elem_t E;
  __shared__ uint32_t h[kMaxRanks][256];
  __shared__ uint32_t scan[256];
  __shared__ uint32_t s_prefix[kMaxRanks], s_rank[kMaxRanks];
  __shared__ uint32_t bucket, rest;
  const uint64_t row = blockIdx.x;
  const E* xr = static_cast<const E*>(x) + row * n;
  if (threadIdx.x < kMaxRanks) {
    s_prefix[threadIdx.x] = 0u;
    s_rank[threadIdx.x] = threadIdx.x < m ? (uint32_t)ranks.r[threadIdx.x] : 0u;
  }
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (uint32_t j = 0; j < m; ++j) h[j][threadIdx.x] = 0;
    __syncthreads();
    uint32_t prefix[kMaxRanks];
    for (uint32_t j = 0; j < kMaxRanks; ++j) prefix[j] = s_prefix[j];
    const uint32_t mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
    for (uint64_t i = threadIdx.x; i < n; i += 256) {
      const uint32_t k = ostat_key(Store<DT>::load1(xr + i));
      const uint32_t b = (k >> shift) & 255u;
#pragma unroll
      for (uint32_t j = 0; j < kMaxRanks; ++j)
        if (j < m && (k & mask) == prefix[j]) atomicAdd(&h[j][b], 1u);
    }
    __syncthreads();
    for (uint32_t j = 0; j < m; ++j) {
      ostat_pick(h[j][threadIdx.x], s_rank[j], scan, &bucket, &rest);
      if (threadIdx.x == 0) {
        s_prefix[j] |= bucket << shift;
        s_rank[j] = rest;
      }
      __syncthreads();
    }
  }
  if (threadIdx.x < m) out[row * m + threadIdx.x] = ostat_value(s_prefix[threadIdx.x]);
}

static size_t ostat_ws_bytes(uint64_t rows, uint32_t m) {
  // [state rows*m, 256-byte padded] [4 histograms rows*m*256] -- long-row mode only
  return rows * m * sizeof(OstatState) + 4ull * rows * m * 256 * sizeof(uint32_t) + 512;
}
constexpr uint64_t kLongRowMaxRows = 64;        // more rows than this: one block per row

template <int DT>
static int launch_ostat(const void* x, uint64_t rows, uint64_t n, const OstatRanks& ranks_dev, uint32_t m, float* out, char* ws,
                        hipStream_t st) {
  if (rows > kLongRowMaxRows || n <= 4096) {
    hipLaunchKernelGGL((ostat_rows_k<DT>), dim3((unsigned)rows), dim3(256), 0, st, x, n, m, ranks_dev, out);
    return check_launch("ostat_rows_k");
  }
  OstatState* state = reinterpret_cast<OstatState*>(ws);
  uint32_t* hist = reinterpret_cast<uint32_t*>(ws + ((rows * m * sizeof(OstatState) + 255) / 256) * 256);
  const size_t hist_pass = rows * m * 256 * sizeof(uint32_t);
  if (hipMemsetAsync(hist, 0, 4 * hist_pass, st) != hipSuccess) return set_error(TQ_ELAUNCH, "tq_order_stats: memset failed");
  hipLaunchKernelGGL(ostat_init_k, dim3((unsigned)ceil_div(rows * m, 256)), dim3(256), 0, st, state, ranks_dev, rows, m);
  const unsigned bpr = (unsigned)std::min<uint64_t>(std::max<uint64_t>(ceil_div(n, 256 * 16), 1), 2048 / std::max<uint64_t>(rows, 1) + 1);
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    uint32_t* hp = hist + (size_t)pass * rows * m * 256;
    hipLaunchKernelGGL((ostat_hist_k<DT>), dim3(bpr, (unsigned)rows), dim3(256), 0, st, x, n, m, shift, state, hp);
    hipLaunchKernelGGL(ostat_select_k, dim3((unsigned)rows), dim3(256), 0, st, hp, state, m, shift, out);
  }
  return check_launch("ostat_hist_k / ostat_select_k");
}

}  // namespace tq

using namespace tq;

extern "C" size_t tq_order_stats_workspace_bytes(uint64_t rows, uint32_t m) {
  return ostat_ws_bytes(std::min<uint64_t>(rows, kLongRowMaxRows), std::min<uint32_t>(m, kMaxRanks));
}

extern "C" int tq_order_stats(const void* x, uint64_t rows, uint64_t n, int dtype, const uint64_t* ranks, uint32_t m, float* out,
                              void* workspace, size_t workspace_bytes, tq_stream_t stream) {
  TQ_REQUIRE(x && ranks && out, "tq_order_stats: NULL pointer");
  TQ_REQUIRE(rows >= 1 && n >= 1, "tq_order_stats: empty input");
  TQ_REQUIRE(n < (1ull << 32), "tq_order_stats: rows of 2^32 or more elements are not supported");
  TQ_REQUIRE(m >= 1 && m <= (uint32_t)kMaxRanks, "tq_order_stats: 1..%d ranks per call, got %u", kMaxRanks, m);
  TQ_REQUIRE(dtype == TQ_F32 || dtype == TQ_BF16 || dtype == TQ_F16, "tq_order_stats: bad dtype %d", dtype);
  for (uint32_t j = 0; j < m; ++j) TQ_REQUIRE(ranks[j] < n, "tq_order_stats: rank %llu outside a row of %llu elements",
                                              (unsigned long long)ranks[j], (unsigned long long)n);
  TQ_REQUIRE(workspace && workspace_bytes >= tq_order_stats_workspace_bytes(rows, m), "tq_order_stats: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  OstatRanks ranks_dev = {{0, 0, 0, 0}};
  for (uint32_t j = 0; j < m; ++j) ranks_dev.r[j] = ranks[j];
  switch (dtype) {
    case TQ_F32: return launch_ostat<TQ_F32>(x, rows, n, ranks_dev, m, out, ws, st);
    case TQ_BF16: return launch_ostat<TQ_BF16>(x, rows, n, ranks_dev, m, out, ws, st);
    default: return launch_ostat<TQ_F16>(x, rows, n, ranks_dev, m, out, ws, st);
  }
}
