// K4/K5 + estimator state + range->parameter kernels for gfx950.
//
//   mm_cols   : statistics per index of the LAST axis (per-embedding).  No transpose copy (the
//               reference does transpose+contiguous+view+min/max, range_estimators.py:82-85,
//               114-116): a 2-D block owns a fixed set of 16-byte column vectors and walks down the
//               rows keeping 8 (bf16) / 4 (fp32) running minima and maxima per lane in registers.
//   mm_rows   : statistics per contiguous row (per-tensor = one row, per-channel = dim 0, any
//               other axis); wave __shfl_xor reduction, LDS across the 4 waves.
//   mm_final  : reduces the block partials ([P][2][n_params] in the caller's workspace).
//   Deterministic: no atomics anywhere.
#include <algorithm>

#include "tq_device.h"
#include "tq_host.h"

namespace tq {

// ------------------------------------------------------------------------------ last axis
// blockDim = (CX, RY): CX lanes side by side cover CX 16-byte vectors of a row, RY rows at a time.
// grid = (col_chunks, row_blocks).  partial layout: ws[(row_block) * 2 * d + {0,1} * d + col]
template <int DT>
__global__ void mm_cols(const u32x4* __restrict__ x, uint64_t rows, uint32_t d, float* __restrict__ ws) {
  constexpr int V = Store<DT>::kVec;
  extern __shared__ __attribute__((aligned(16))) float s_mm[];   // [RY][2][CX*V]
  const uint32_t vpr = d / V;
  const uint32_t cx = blockIdx.x * blockDim.x + threadIdx.x;    // vector column
  const bool live = cx < vpr;
  MinMax acc[V];
  float mn[V], mx[V];

  // block (., by) owns the contiguous rows [by * chunk, (by + 1) * chunk): the resident blocks
  // then sweep one contiguous window of HBM instead of gridDim.y windows that lie MBs apart
  const uint64_t chunk = (rows + gridDim.y - 1) / gridDim.y;
  const uint64_t row_end = min(rows, ((uint64_t)blockIdx.y + 1) * chunk);
  const uint64_t row_stride = blockDim.y;
  uint64_t r = (uint64_t)blockIdx.y * chunk + threadIdx.y;
  if (live) {
    constexpr int U = 4;
    for (; r + (U - 1) * row_stride < row_end; r += U * row_stride) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = ld_stream(x + (r + u * row_stride) * vpr + cx);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float f[V];
        Store<DT>::unpack(v[u], f);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j].add(f[j]);
      }
    }
    for (; r < row_end; r += row_stride) {
      float f[V];
      Store<DT>::unpack(x[r * vpr + cx], f);
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j].add(f[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) { mn[j] = acc[j].lo(); mx[j] = acc[j].hi(); }
  // combine the RY row-lanes of this block through LDS
  const uint32_t cw = blockDim.x * V;
  float* my = s_mm + (size_t)threadIdx.y * 2 * cw;
#pragma unroll
  for (int j = 0; j < V; ++j) { my[threadIdx.x * V + j] = mn[j]; my[cw + threadIdx.x * V + j] = mx[j]; }
  __syncthreads();
  if (threadIdx.y == 0 && live) {
    for (uint32_t k = 1; k < blockDim.y; ++k) {
      const float* o = s_mm + (size_t)k * 2 * cw;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        mn[j] = min_nanprop(mn[j], o[threadIdx.x * V + j]);
        mx[j] = max_nanprop(mx[j], o[cw + threadIdx.x * V + j]);
      }
    }
    float* out = ws + (size_t)blockIdx.y * 2 * d;
#pragma unroll
    for (int j = 0; j < V; ++j) { out[cx * V + j] = mn[j]; out[d + cx * V + j] = mx[j]; }
  }
}

// ------------------------------------------------------------------------------ rows
// x viewed as [n_rows, inner]; parameter = row % n_params; grid = (chunks, rows_in_grid).
// partial layout: ws[((row / n_params) * chunks + chunk) * 2 * n_params + {0,1} * n_params + param]
template <int DT, bool VEC>
__global__ __launch_bounds__(kBlock) void mm_rows(const void* __restrict__ x, uint64_t n_rows, uint64_t inner,
                                                  uint64_t n_params, float* __restrict__ ws) {
  constexpr int V = Store<DT>::kVec;
  typedef typename Store<DT>::elem_t E;
  __shared__ float s_red[2][kBlock / kWave];
  for (uint64_t row = blockIdx.y; row < n_rows; row += gridDim.y) {
    MinMax acc;
    const uint64_t stride = kBlock;
    if (VEC) {
      // block bx owns the contiguous vectors [bx * chunk, (bx + 1) * chunk) of the row
      const u32x4* xv = static_cast<const u32x4*>(x) + row * (inner / V);
      const uint64_t n_all = inner / V;
      const uint64_t chunk = (n_all + gridDim.x - 1) / gridDim.x;
      const uint64_t n_vec = min(n_all, ((uint64_t)blockIdx.x + 1) * chunk);
      constexpr int U = 4;
      uint64_t i = (uint64_t)blockIdx.x * chunk + threadIdx.x;
      for (; i + (U - 1) * stride < n_vec; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld_stream(xv + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          float f[V];
          Store<DT>::unpack(v[u], f);
#pragma unroll
          for (int j = 0; j < V; ++j) acc.add(f[j]);
        }
      }
      for (; i < n_vec; i += stride) {
        float f[V];
        Store<DT>::unpack(xv[i], f);
#pragma unroll
        for (int j = 0; j < V; ++j) acc.add(f[j]);
      }
    } else {
      const E* xs = static_cast<const E*>(x) + row * inner;
      for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < inner; i += (uint64_t)gridDim.x * kBlock)
        acc.add(Store<DT>::load1(xs + i));
    }
    float mn = wave_min(acc.lo());
    float mx = wave_max(acc.hi());
    const int w = threadIdx.x / kWave;
    __syncthreads();   // s_red reuse across row iterations
    if ((threadIdx.x & (kWave - 1)) == 0) { s_red[0][w] = mn; s_red[1][w] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int k = 1; k < kBlock / kWave; ++k) { mn = min_nanprop(mn, s_red[0][k]); mx = max_nanprop(mx, s_red[1][k]); }
      const uint64_t outer = row / n_params, p = row % n_params;
      float* out = ws + ((outer * gridDim.x) + blockIdx.x) * 2 * n_params;
      out[p] = mn;
      out[n_params + p] = mx;
    }
  }
}

// ------------------------------------------------------------------------------ short rows, one wave per (parameter, slice)
// x viewed as [outer, n_params, inner] with rows of at most a few KB (per-token ranges of a [B, T, d] activation:
// n_params = T, inner = d, reference main.py:359-376 `--per-token`; per-channel weights).  mm_rows gives such a row a
// whole 256-thread block (96 of 256 lanes loading for d = 768 bf16, two barriers per row).  Here wave (p, s) owns the
// rows o = s, s + S, s + 2 S, ... of parameter p, keeps lane-local running statistics over ALL of them and reduces
// across the wave once at the end; consecutive waves own consecutive parameters of the same `o`, so that at any time the
// resident waves sweep one contiguous window of S x n_params rows.  No LDS, no barrier.
// partial layout (mm_final's): ws[s * 2 * n_params + {0,1} * n_params + p]
template <int DT>
__global__ __launch_bounds__(kBlock) void mm_rows_wave(const u32x4* __restrict__ x, uint64_t outer, uint32_t vpr,
                                                       uint32_t n_params, uint32_t S, float* __restrict__ ws) {
  constexpr int V = Store<DT>::kVec;
  constexpr int U = 4;
  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint64_t gw = (uint64_t)blockIdx.x * (kBlock / kWave) + __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  if (gw >= (uint64_t)n_params * S) return;
  const uint32_t p = (uint32_t)(gw % n_params), s = (uint32_t)(gw / n_params);
  const uint64_t row_stride = (uint64_t)n_params * vpr;         // vectors between two rows of one parameter
  const u32x4* base = x + (uint64_t)p * vpr;
  MinMax acc;
  uint64_t o = s;
  // Rows that are not a whole number of wave widths (d = 768 in 16-bit storage: 96 vectors, the second pass over a row had
  // 32 of 64 lanes loading: 75 % of HBM against 86 % for fp32's 192 vectors): the U rows of a batch as ONE index space of
  // U * vpr vectors -- 6 full passes instead of 8 half-empty ones.  A lane steps its (row, offset) pair by the wave width:
  // at most one row boundary per step since vpr > 64; no division.  (Round 5 tried the same for the fake-quant kernel and
  // found it slower; for the statistics, which do a quarter of the arithmetic per vector, it pays -- profiles/r06/mm_flat_ab.txt.)
  if (vpr > kWave && vpr % kWave != 0) {
    const uint64_t rs = (uint64_t)S * row_stride;                     // vectors between two rows of a batch
    const uint32_t total = U * vpr;
    for (; o + (uint64_t)(U - 1) * S < outer; o += (uint64_t)U * S) {
      uint32_t i = lane;
      uint64_t at = o * row_stride + lane;
      for (uint32_t g0 = 0; g0 < total; g0 += U * kWave) {
        u32x4 v[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          ok[u] = g0 + (uint32_t)u * kWave + lane < total;
          if (ok[u]) v[u] = ld_stream(base + at);
          i += kWave;
          at += kWave;
          if (i >= vpr) { i -= vpr; at += rs - vpr; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (ok[u]) {
            float f[V];
            Store<DT>::unpack(v[u], f);
#pragma unroll
            for (int j = 0; j < V; ++j) acc.add(f[j]);
          }
        }
      }
    }
  }
  for (; o + (uint64_t)(U - 1) * S < outer; o += (uint64_t)U * S) {
    for (uint32_t i = lane; i < vpr; i += kWave) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = ld_stream(base + (o + (uint64_t)u * S) * row_stride + i);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float f[V];
        Store<DT>::unpack(v[u], f);
#pragma unroll
        for (int j = 0; j < V; ++j) acc.add(f[j]);
      }
    }
  }
  for (; o < outer; o += S) {
    for (uint32_t i = lane; i < vpr; i += kWave) {
      float f[V];
      Store<DT>::unpack(ld_stream(base + o * row_stride + i), f);
#pragma unroll
      for (int j = 0; j < V; ++j) acc.add(f[j]);
    }
  }
  const float mn = wave_min(acc.lo()), mx = wave_max(acc.hi());
  if (lane == 0) {
    float* out = ws + (uint64_t)s * 2 * n_params;
    out[p] = mn;
    out[n_params + p] = mx;
  }
}

// ------------------------------------------------------------------------------ finalize
// ws [P][2][n_params] -> out_min[n_params], out_max[n_params].  block = (cx, sy): cx adjacent
// parameters, sy slices of P.
// neg_min: store -min (the [-min ; max] layout of the sharded-calibration exchange buffer, tq_calibrate_stats).
__global__ void mm_final(const float* __restrict__ ws, uint64_t P, uint64_t n_params, float* __restrict__ out_min,
                         float* __restrict__ out_max, bool neg_min) {
  extern __shared__ float s_f[];   // [2][sy][cx]
  const uint32_t cx = blockDim.x, sy = blockDim.y;
  const uint64_t col = (uint64_t)blockIdx.x * cx + threadIdx.x;
  float mn = kInf, mx = -kInf;
  if (col < n_params) {
    // 4 independent loads in flight per lane: a single dependent chain over P records is
    // latency-bound (33 us for 16384 records before this unroll)
    uint64_t p = threadIdx.y;
    for (; p + 3 * (uint64_t)sy < P; p += 4 * (uint64_t)sy) {
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = ws[(p + u * sy) * 2 * n_params + col];
        b[u] = ws[(p + u * sy) * 2 * n_params + n_params + col];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { mn = min_nanprop(mn, a[u]); mx = max_nanprop(mx, b[u]); }
    }
    for (; p < P; p += sy) {
      mn = min_nanprop(mn, ws[p * 2 * n_params + col]);
      mx = max_nanprop(mx, ws[p * 2 * n_params + n_params + col]);
    }
  }
  s_f[threadIdx.y * cx + threadIdx.x] = mn;
  s_f[(sy + threadIdx.y) * cx + threadIdx.x] = mx;
  __syncthreads();
  // tree over the sy slices
  for (uint32_t h = 1; h < sy; h <<= 1) {
    if ((threadIdx.y % (2 * h)) == 0 && threadIdx.y + h < sy) {
      float* a = s_f + threadIdx.y * cx + threadIdx.x;
      float* b = s_f + (sy + threadIdx.y) * cx + threadIdx.x;
      *a = min_nanprop(*a, a[h * cx]);
      *b = max_nanprop(*b, b[h * cx]);
    }
    __syncthreads();
  }
  if (threadIdx.y == 0 && col < n_params) {
    out_min[col] = neg_min ? -s_f[threadIdx.x] : s_f[threadIdx.x];
    out_max[col] = s_f[sy * cx + threadIdx.x];
  }
}

struct MMPlan {
  bool cols;          // mm_cols path
  bool wave;          // mm_rows_wave path (S = P slices of the outer index)
  unsigned bx, by, gx, gy;
  uint64_t P;         // number of partial records of 2*n_params floats
};

static MMPlan plan_minmax(uint64_t n, uint64_t n_params, uint64_t inner, int V, bool aligned) {
  MMPlan pl{};
  if (n_params > 1 && inner == 1 && n_params % V == 0 && aligned) {
    const uint64_t vpr = n_params / V, rows = n / n_params;
    pl.cols = true;
    pl.bx = (unsigned)std::min<uint64_t>(vpr, 128);
    pl.by = std::max(1u, 256u / pl.bx);
    pl.gx = (unsigned)ceil_div(vpr, pl.bx);
    // enough row-blocks to fill the chip, each walking >= 8 rows per lane
    const uint64_t want = std::max<uint64_t>(1, (uint64_t)kMaxGrid / 2 / pl.gx);
    pl.gy = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(want, ceil_div(rows, (uint64_t)pl.by * 8)));
    pl.P = pl.gy;
    return pl;
  }
  if (n_params > 1 && inner > 1 && inner % V == 0 && aligned && inner / V <= kWaveRowMaxVec && n_params <= (1u << 24) &&
      n % (n_params * inner) == 0) {
    const uint64_t outer = n / (n_params * inner);
    pl.wave = true;
    pl.P = std::max<uint64_t>(1, std::min<uint64_t>(ceil_div(kWaveRowTarget, n_params), outer));
    pl.gx = (unsigned)ceil_div(n_params * pl.P, (uint64_t)(kBlock / kWave));
    return pl;
  }
  const uint64_t eff_inner = n_params == 1 ? n : inner;
  const uint64_t n_rows = n_params == 1 ? 1 : n / inner;
  const uint64_t per_block = (uint64_t)kBlock * V * 4 * 2;      // two 4-vector rounds per lane
  pl.cols = false;
  pl.gy = (unsigned)std::min<uint64_t>(n_rows, 65535);
  constexpr uint64_t kMaxBlocks = 16384;                        // partial records stay <= 128 KiB
  const uint64_t max_gx = std::max<uint64_t>(1, kMaxBlocks / std::max<uint64_t>(1, std::min<uint64_t>(n_rows, kMaxBlocks)));
  pl.gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ceil_div(eff_inner, per_block), max_gx));
  const uint64_t outer = n_params == 1 ? 1 : n_rows / n_params;
  pl.P = outer * pl.gx;
  return pl;
}

template <int DT>
static int launch_minmax(const void* x, uint64_t n, uint64_t n_params, uint64_t inner, float* out_min,
                         float* out_max, float* ws, size_t ws_bytes, hipStream_t st, bool neg_min = false) {
  constexpr int V = Store<DT>::kVec;
  const bool al = aligned16(x);
  const MMPlan pl = plan_minmax(n, n_params, inner, V, al);
  const size_t need = pl.P * 2 * n_params * sizeof(float);
  if (ws == nullptr || ws_bytes < need)
    return set_error(TQ_EWORKSPACE, "tq_minmax: workspace %zu < %zu bytes", ws_bytes, need);
  if (pl.cols) {
    const size_t lds = (size_t)pl.by * 2 * pl.bx * V * sizeof(float);
    hipLaunchKernelGGL((mm_cols<DT>), dim3(pl.gx, pl.gy), dim3(pl.bx, pl.by), lds, st,
                       static_cast<const u32x4*>(x), n / n_params, (uint32_t)n_params, ws);
  } else if (pl.wave) {
    hipLaunchKernelGGL((mm_rows_wave<DT>), dim3(pl.gx), dim3(kBlock), 0, st, static_cast<const u32x4*>(x),
                       n / (n_params * inner), (uint32_t)(inner / V), (uint32_t)n_params, (uint32_t)pl.P, ws);
  } else {
    const uint64_t eff_inner = n_params == 1 ? n : inner;
    const uint64_t n_rows = n_params == 1 ? 1 : n / inner;
    const bool vec = al && (eff_inner % V == 0);
    if (vec) hipLaunchKernelGGL((mm_rows<DT, true>), dim3(pl.gx, pl.gy), dim3(kBlock), 0, st, x, n_rows, eff_inner, n_params, ws);
    else     hipLaunchKernelGGL((mm_rows<DT, false>), dim3(pl.gx, pl.gy), dim3(kBlock), 0, st, x, n_rows, eff_inner, n_params, ws);
  }
  if (int e = check_launch("tq_minmax partial")) return e;
  const unsigned cx = (unsigned)std::min<uint64_t>(n_params, 64);
  const unsigned sy = std::max(1u, std::min<unsigned>(1024 / cx, (unsigned)std::max<uint64_t>(1, pl.P)));
  hipLaunchKernelGGL(mm_final, dim3((unsigned)ceil_div(n_params, cx)), dim3(cx, sy), 2 * sy * cx * sizeof(float), st,
                     ws, pl.P, n_params, out_min, out_max, neg_min);
  return check_launch("tq_minmax final");
}

// ------------------------------------------------------------------------------ estimator state
// One block.  Step 1 (optional): fold per-dimension statistics into per-group statistics
// (range_estimators.py:87-112 / :183-193).  Step 2: current / all / running update.
__global__ void range_update_k(int mode, const float* __restrict__ new_min, const float* __restrict__ new_max,
                               float* __restrict__ cur_min, float* __restrict__ cur_max, uint64_t n, int initialised,
                               double momentum_d, uint64_t n_groups, const int64_t* __restrict__ order) {
  extern __shared__ float s_g[];   // [2][n_groups]
  if (n_groups > 0) {
    const uint64_t gs = n / n_groups;
    for (uint64_t g = threadIdx.x; g < n_groups; g += blockDim.x) {
      float mn = kInf, mx = -kInf;
      for (uint64_t k = 0; k < gs; ++k) {
        const uint64_t dim = order ? (uint64_t)order[g * gs + k] : g * gs + k;
        mn = min_nanprop(mn, new_min[dim]);
        mx = max_nanprop(mx, new_max[dim]);
      }
      s_g[g] = mn;
      s_g[n_groups + g] = mx;
    }
    __syncthreads();
  }
  // (1 - momentum) as the reference evaluates it: python double, narrowed by the fp32 multiply
  const float om = (float)(1.0 - momentum_d);
  const float momentum = (float)momentum_d;
  for (uint64_t j = threadIdx.x; j < n; j += blockDim.x) {
    float a, b;
    if (n_groups > 0) {
      // position j of the (permuted) layout belongs to group j / gs and is dimension order[j]
      const uint64_t gs = n / n_groups;
      const uint64_t dim = order ? (uint64_t)order[j] : j;
      a = s_g[j / gs];
      b = s_g[n_groups + j / gs];
      // writes go to `dim`
      if (mode == TQ_EST_CURRENT || !initialised) { cur_min[dim] = a; cur_max[dim] = b; }
      else if (mode == TQ_EST_ALL) { cur_min[dim] = min_nanprop(cur_min[dim], a); cur_max[dim] = max_nanprop(cur_max[dim], b); }
      else { cur_min[dim] = om * a + momentum * cur_min[dim]; cur_max[dim] = om * b + momentum * cur_max[dim]; }
      continue;
    }
    a = new_min[j];
    b = new_max[j];
    if (mode == TQ_EST_CURRENT || !initialised) { cur_min[j] = a; cur_max[j] = b; }
    else if (mode == TQ_EST_ALL) { cur_min[j] = min_nanprop(cur_min[j], a); cur_max[j] = max_nanprop(cur_max[j], b); }
    else { cur_min[j] = om * a + momentum * cur_min[j]; cur_max[j] = om * b + momentum * cur_max[j]; }
  }
}

__global__ void axis_ranges_k(const float* __restrict__ new_min, const float* __restrict__ new_max,
                              float* __restrict__ ranges, uint64_t n, int first) {
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
    const float r = new_max[j] - new_min[j];
    // range_estimators.py:75-79: momentum * ranges + (1 - momentum) * ranges with momentum = 0.1
    ranges[j] = first ? r : (0.1f * r + 0.9f * r);
  }
}

// ------------------------------------------------------------------------------ range -> params
__global__ void set_range_asym_k(const float* __restrict__ x_min, const float* __restrict__ x_max, uint64_t n,
                                 int n_bits, float eps, int log_domain, float* __restrict__ delta,
                                 float* __restrict__ zero_float) {
  const float top = grid_top(n_bits);
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
    const float lo = min_nanprop(x_min[j], 0.0f);       // quantizers.py:258
    const float hi = max_nanprop(x_max[j], eps);        // :259  (ones_like * eps == eps in fp32)
    const float d = (hi - lo) / top;                    // :276
    zero_float[j] = (-lo) / d;                          // :277
    delta[j] = log_domain ? logf(d) : d;                // :279-280
  }
}

// single block: `signed` is a property of the whole range vector (quantizers.py:336)
__global__ void set_range_sym_k(const float* __restrict__ x_min, const float* __restrict__ x_max, uint64_t n,
                                int n_bits, float eps, int log_domain, float* __restrict__ delta,
                                uint8_t* __restrict__ signed_flag) {
  __shared__ int s_neg;
  if (threadIdx.x == 0) s_neg = 0;
  __syncthreads();
  int neg = 0;
  for (uint64_t j = threadIdx.x; j < n; j += blockDim.x) neg |= (min_nanprop(x_min[j], 0.0f) < 0.0f) ? 1 : 0;
  if (neg) atomicOr(&s_neg, 1);
  __syncthreads();
  const bool sgn = s_neg != 0;
  if (threadIdx.x == 0) signed_flag[0] = sgn ? 1 : 0;
  const float top = grid_top(n_bits - (sgn ? 1 : 0));   // :325-328
  for (uint64_t j = threadIdx.x; j < n; j += blockDim.x) {
    const float lo = min_nanprop(x_min[j], 0.0f);
    const float hi = max_nanprop(x_max[j], eps);
    const float d = max_nanprop(fabsf(lo), hi) / top;   // :338-339
    delta[j] = log_domain ? logf(d) : d;
  }
}

// ------------------------------------------------------------------------------ fused calibration step
// group fold + estimator update + range -> quantizer parameters in ONE single-block launch
// (range_update_k followed by set_range_*_k); n <= kCalibMaxN so the new state fits in LDS.
constexpr uint32_t kCalibMaxN = 4096;

// prev_* may alias cur_* (in-place state): every thread reads prev[dim] before it writes cur[dim].
__global__ __launch_bounds__(1024) void calib_update_k(int mode, const float* __restrict__ new_min,
                                                       const float* __restrict__ new_max, const float* prev_min,
                                                       const float* prev_max, float* cur_min,
                                                       float* cur_max, uint32_t n, double momentum_d,
                                                       uint32_t n_groups, const int64_t* __restrict__ order, int n_bits,
                                                       int symmetric, float eps, int log_domain, float* __restrict__ delta,
                                                       float* __restrict__ zero_float, uint8_t* __restrict__ signed_flag,
                                                       int neg_min /* new_min holds -min (exchange-buffer layout) */) {
  extern __shared__ float s_c[];          // [2][n] new state, then [2][n_groups]
  float* s_lo = s_c;
  float* s_hi = s_c + n;
  float* s_g = s_c + 2 * n;
  __shared__ int s_neg;
  if (threadIdx.x == 0) s_neg = 0;
  if (n_groups > 0) {
    const uint32_t gs = n / n_groups;
    for (uint32_t g = threadIdx.x; g < n_groups; g += blockDim.x) {
      float mn = kInf, mx = -kInf;
      for (uint32_t k = 0; k < gs; ++k) {
        const uint32_t dim = order ? (uint32_t)order[g * gs + k] : g * gs + k;
        mn = min_nanprop(mn, neg_min ? -new_min[dim] : new_min[dim]);
        mx = max_nanprop(mx, new_max[dim]);
      }
      s_g[g] = mn;
      s_g[n_groups + g] = mx;
    }
  }
  __syncthreads();
  const bool first = prev_min == nullptr;
  const float om = (float)(1.0 - momentum_d), mom = (float)momentum_d;
  for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
    uint32_t dim = j;
    float a, b;
    if (n_groups > 0) {
      const uint32_t gs = n / n_groups;
      dim = order ? (uint32_t)order[j] : j;
      a = s_g[j / gs];
      b = s_g[n_groups + j / gs];
    } else {
      a = neg_min ? -new_min[j] : new_min[j];
      b = new_max[j];
    }
    if (!(mode == TQ_EST_CURRENT || first)) {
      const float pa = prev_min[dim], pb = prev_max[dim];
      if (mode == TQ_EST_ALL) { a = min_nanprop(pa, a); b = max_nanprop(pb, b); }
      else { a = om * a + mom * pa; b = om * b + mom * pb; }
    }
    cur_min[dim] = a;
    cur_max[dim] = b;
    s_lo[dim] = min_nanprop(a, 0.0f);            // quantizers.py:258
    s_hi[dim] = max_nanprop(b, eps);             // :259
  }
  __syncthreads();
  if (symmetric) {
    int neg = 0;
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) neg |= s_lo[j] < 0.0f ? 1 : 0;
    if (neg) atomicOr(&s_neg, 1);
    __syncthreads();
    const bool sgn = s_neg != 0;
    if (threadIdx.x == 0) signed_flag[0] = sgn ? 1 : 0;
    const float top = grid_top(n_bits - (sgn ? 1 : 0));
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
      const float d = max_nanprop(fabsf(s_lo[j]), s_hi[j]) / top;
      delta[j] = log_domain ? logf(d) : d;
    }
  } else {
    const float top = grid_top(n_bits);
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
      const float d = (s_hi[j] - s_lo[j]) / top;
      zero_float[j] = (-s_lo[j]) / d;
      delta[j] = log_domain ? logf(d) : d;
    }
  }
}

// ------------------------------------------------------------------------------ per-tensor calibration, one launch
// Statistics + estimator update + range -> parameters for ONE range (per-tensor quantizers: 161 of the
// 161 activation sites of the BERT W8A8 config).  Every block reduces its contiguous chunk to a partial
// (as mm_rows does) and takes a ticket; the block that draws the last ticket reduces the partials
// (min / max are order-independent, so the result does not depend on which block that is), applies the
// (Round 3: with a quantized output requested the step is ticket-free instead -- calib_partials_k below leaves the block
// partials, the quantizer launch folds them in every block: 8.3 + 3.8 us -> 3.7 + 4.3 us at [8, 128, 768].)
// estimator rule and writes state and quantizer parameters.  2 dependent launches per calibrating
// call (this + the quantizer) instead of 4.  `counter` must be 0 on entry and is 0 again on exit.
template <int DT, bool VEC>
__global__ __launch_bounds__(kBlock) void calib_tensor_k(const void* __restrict__ x, uint64_t n, float* ws,
                                                         uint32_t* counter, int mode, const float* prev_min,
                                                         const float* prev_max, float* cur_min, float* cur_max,
                                                         double momentum_d, int n_bits, int symmetric, float eps,
                                                         int log_domain, float* delta, float* zero_float,
                                                         uint8_t* signed_flag,
                                                         float* stats_out /* non-NULL: only write [-min, max] */) {
  constexpr int V = Store<DT>::kVec;
  typedef typename Store<DT>::elem_t E;
  __shared__ float s_red[2][kBlock / kWave];
  __shared__ uint32_t s_ticket;
  MinMax acc;
  if (VEC) {
    const u32x4* xv = static_cast<const u32x4*>(x);
    const uint64_t n_all = n / V;
    const uint64_t chunk = (n_all + gridDim.x - 1) / gridDim.x;
    const uint64_t n_vec = min(n_all, ((uint64_t)blockIdx.x + 1) * chunk);
    constexpr int U = 4;
    uint64_t i = (uint64_t)blockIdx.x * chunk + threadIdx.x;
    for (; i + (U - 1) * (uint64_t)kBlock < n_vec; i += U * (uint64_t)kBlock) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = ld_stream(xv + i + u * (uint64_t)kBlock);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float f[V];
        Store<DT>::unpack(v[u], f);
#pragma unroll
        for (int j = 0; j < V; ++j) acc.add(f[j]);
      }
    }
    for (; i < n_vec; i += kBlock) {
      float f[V];
      Store<DT>::unpack(xv[i], f);
#pragma unroll
      for (int j = 0; j < V; ++j) acc.add(f[j]);
    }
    if (blockIdx.x == 0 && n_all * V + threadIdx.x < n)                      // ragged tail (< V elements)
      acc.add(Store<DT>::load1(static_cast<const E*>(x) + n_all * V + threadIdx.x));
  } else {
    const E* xs = static_cast<const E*>(x);
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock)
      acc.add(Store<DT>::load1(xs + i));
  }
  const int w = threadIdx.x / kWave;
  float mn = wave_min(acc.lo()), mx = wave_max(acc.hi());
  if ((threadIdx.x & (kWave - 1)) == 0) { s_red[0][w] = mn; s_red[1][w] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < kBlock / kWave; ++k) { mn = min_nanprop(mn, s_red[0][k]); mx = max_nanprop(mx, s_red[1][k]); }
    __hip_atomic_store(ws + 2 * blockIdx.x, mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(ws + 2 * blockIdx.x + 1, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    s_ticket = atomicAdd(counter, 1u);
  }
  __syncthreads();
  if (s_ticket != gridDim.x - 1) return;

  // ---- last block: partials -> statistic -> estimator state -> quantizer parameters -------------------------
  __threadfence();
  mn = kInf; mx = -kInf;
  {
    // the fence invalidated this CU's L1 (buffer_inv sc1), so plain loads see every block's partial; 4 independent
    // loads in flight per lane (atomic / volatile loads are issued one at a time, ~1.5 us each)
    const float* wsv = ws;
    uint32_t b = threadIdx.x;
    for (; b + 3 * kBlock < gridDim.x; b += 4 * kBlock) {
      float lo4[4], hi4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { lo4[u] = wsv[2 * (b + u * kBlock)]; hi4[u] = wsv[2 * (b + u * kBlock) + 1]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { mn = min_nanprop(mn, lo4[u]); mx = max_nanprop(mx, hi4[u]); }
    }
    for (; b < gridDim.x; b += kBlock) { mn = min_nanprop(mn, wsv[2 * b]); mx = max_nanprop(mx, wsv[2 * b + 1]); }
  }
  mn = wave_min(mn); mx = wave_max(mx);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) { s_red[0][w] = mn; s_red[1][w] = mx; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (int k = 1; k < kBlock / kWave; ++k) { mn = min_nanprop(mn, s_red[0][k]); mx = max_nanprop(mx, s_red[1][k]); }
  if (stats_out != nullptr) {            // sharded calibration: the exchange happens between this and the update
    stats_out[0] = -mn;
    stats_out[1] = mx;
    if (prev_min != nullptr) {           // one-call sharded steps: a copy of the previous state behind the statistics, so
      stats_out[2] = prev_min[0];        // that the quantizer launch (whose block 0 stores the new state, possibly in
      stats_out[3] = prev_max[0];        // place) never reads the live buffers
    }
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  float a = mn, b = mx;
  if (!(mode == TQ_EST_CURRENT || prev_min == nullptr)) {
    const float pa = prev_min[0], pb = prev_max[0];
    const float om = (float)(1.0 - momentum_d), mom = (float)momentum_d;
    if (mode == TQ_EST_ALL) { a = min_nanprop(pa, a); b = max_nanprop(pb, b); }     // range_estimators.py:162-167
    else { a = om * a + mom * pa; b = om * b + mom * pb; }                          // :209-214
  }
  cur_min[0] = a;
  cur_max[0] = b;
  const float lo = min_nanprop(a, 0.0f), hi = max_nanprop(b, eps);                  // quantizers.py:258-259
  if (symmetric) {
    const bool sgn = lo < 0.0f;
    signed_flag[0] = sgn ? 1 : 0;
    const float d = max_nanprop(fabsf(lo), hi) / grid_top(n_bits - (sgn ? 1 : 0));  // :334-344
    delta[0] = log_domain ? logf(d) : d;
  } else {
    const float d = (hi - lo) / grid_top(n_bits);                                   // :276-277
    zero_float[0] = (-lo) / d;
    delta[0] = log_domain ? logf(d) : d;
  }
  __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


}  // namespace tq

using namespace tq;

extern "C" size_t tq_minmax_workspace_bytes(uint64_t n, uint64_t n_params, uint64_t inner) {
  if (n == 0 || n_params == 0) return 0;
  // the larger of the plans over dtypes / alignment (vector width 4 or 8)
  size_t best = 0;
  for (int V : {4, 8})
    for (bool al : {true, false}) {
      const MMPlan pl = plan_minmax(n, n_params, n_params > 1 && inner == 0 ? 1 : inner, V, al);
      best = std::max(best, (size_t)(pl.P * 2 * n_params * sizeof(float)));
    }
  return best;
}

extern "C" int tq_minmax(const void* x, uint64_t n, int dtype, uint64_t n_params, uint64_t inner, float* out_min,
                         float* out_max, void* workspace, size_t workspace_bytes, tq_stream_t stream) {
  TQ_REQUIRE(x && out_min && out_max, "tq_minmax: NULL pointer");
  TQ_REQUIRE(n > 0, "tq_minmax: empty tensor has no min/max");
  TQ_REQUIRE(n_params >= 1, "tq_minmax: n_params == 0");
  TQ_REQUIRE(dtype == TQ_F32 || dtype == TQ_BF16 || dtype == TQ_F16, "tq_minmax: bad dtype %d", dtype);
  if (n_params > 1) {
    TQ_REQUIRE(inner >= 1, "tq_minmax: inner == 0");
    TQ_REQUIRE(n % (n_params * inner) == 0, "tq_minmax: n not a multiple of n_params*inner");
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* ws = static_cast<float*>(workspace);
  switch (dtype) {
    case TQ_F32: return launch_minmax<TQ_F32>(x, n, n_params, inner, out_min, out_max, ws, workspace_bytes, st);
    case TQ_BF16: return launch_minmax<TQ_BF16>(x, n, n_params, inner, out_min, out_max, ws, workspace_bytes, st);
    default: return launch_minmax<TQ_F16>(x, n, n_params, inner, out_min, out_max, ws, workspace_bytes, st);
  }
}

extern "C" int tq_range_update(int mode, const float* new_min, const float* new_max, float* cur_min, float* cur_max,
                               uint64_t n, int initialised, double momentum, uint64_t n_groups, const int64_t* order,
                               tq_stream_t stream) {
  TQ_REQUIRE(new_min && new_max && cur_min && cur_max, "tq_range_update: NULL pointer");
  TQ_REQUIRE(mode >= TQ_EST_CURRENT && mode <= TQ_EST_RUNNING, "tq_range_update: bad mode %d", mode);
  TQ_REQUIRE(n > 0, "tq_range_update: n == 0");
  TQ_REQUIRE(n_groups == 0 || n % n_groups == 0, "tq_range_update: n %% n_groups != 0");
  TQ_REQUIRE(n_groups * 2 * sizeof(float) <= 64 * 1024, "tq_range_update: too many groups");
  hipLaunchKernelGGL(range_update_k, dim3(1), dim3(n >= 256 ? 1024 : 256), n_groups * 2 * sizeof(float),
                     static_cast<hipStream_t>(stream), mode, new_min, new_max, cur_min, cur_max, n, initialised,
                     momentum, n_groups, order);
  return check_launch("tq_range_update");
}

extern "C" int tq_axis_ranges(const float* new_min, const float* new_max, float* ranges, uint64_t n, int first,
                              tq_stream_t stream) {
  TQ_REQUIRE(new_min && new_max && ranges && n > 0, "tq_axis_ranges: bad argument");
  hipLaunchKernelGGL(axis_ranges_k, dim3((unsigned)std::min<uint64_t>(ceil_div(n, 256), 1024)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), new_min, new_max, ranges, n, first);
  return check_launch("tq_axis_ranges");
}

extern "C" int tq_set_range_asym(const float* x_min, const float* x_max, uint64_t n, int n_bits, float eps,
                                 int log_domain, float* delta, float* zero_float, tq_stream_t stream) {
  TQ_REQUIRE(x_min && x_max && delta && zero_float && n > 0, "tq_set_range_asym: bad argument");
  TQ_REQUIRE(n_bits >= 1 && n_bits <= 24, "tq_set_range_asym: n_bits=%d", n_bits);
  hipLaunchKernelGGL(set_range_asym_k, dim3((unsigned)std::min<uint64_t>(ceil_div(n, 256), 1024)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x_min, x_max, n, n_bits, eps, log_domain, delta, zero_float);
  return check_launch("tq_set_range_asym");
}

extern "C" int tq_set_range_sym(const float* x_min, const float* x_max, uint64_t n, int n_bits, float eps,
                                int log_domain, float* delta, uint8_t* signed_flag, tq_stream_t stream) {
  TQ_REQUIRE(x_min && x_max && delta && signed_flag && n > 0, "tq_set_range_sym: bad argument");
  TQ_REQUIRE(n_bits >= 1 && n_bits <= 24, "tq_set_range_sym: n_bits=%d", n_bits);
  hipLaunchKernelGGL(set_range_sym_k, dim3(1), dim3(n >= 256 ? 1024 : 256), 0, static_cast<hipStream_t>(stream),
                     x_min, x_max, n, n_bits, eps, log_domain, delta, signed_flag);
  return check_launch("tq_set_range_sym");
}

extern "C" size_t tq_calibrate_workspace_bytes(uint64_t n, uint64_t n_params, uint64_t inner) {
  return tq_minmax_workspace_bytes(n, n_params, inner) + 2 * n_params * sizeof(float) + 256;
}

extern "C" int tq_calibrate_minmax(const void* x, uint64_t n, int dtype, uint64_t n_params, uint64_t inner, int mode,
                                   const float* prev_min, const float* prev_max, float* cur_min, float* cur_max,
                                   double momentum, uint64_t n_groups, const int64_t* order, int n_bits, int symmetric,
                                   float eps, int log_domain, float* delta, float* zero_float, uint8_t* signed_flag,
                                   void* y, void* workspace, size_t workspace_bytes, tq_stream_t stream) {
  TQ_REQUIRE(cur_min && cur_max && delta, "tq_calibrate_minmax: NULL output");
  TQ_REQUIRE((prev_min == nullptr) == (prev_max == nullptr), "tq_calibrate_minmax: prev_min / prev_max mismatch");
  TQ_REQUIRE(symmetric ? signed_flag != nullptr : zero_float != nullptr, "tq_calibrate_minmax: missing parameter output");
  TQ_REQUIRE(mode >= TQ_EST_CURRENT && mode <= TQ_EST_RUNNING, "tq_calibrate_minmax: bad mode %d", mode);
  TQ_REQUIRE(n_params >= 1 && n_params <= kCalibMaxN, "tq_calibrate_minmax: n_params=%llu > %u (use the separate calls)",
             (unsigned long long)n_params, kCalibMaxN);
  TQ_REQUIRE(n_groups == 0 || n_params % n_groups == 0, "tq_calibrate_minmax: n_params %% n_groups != 0");
  TQ_REQUIRE(n_bits >= 1 && n_bits <= 24, "tq_calibrate_minmax: n_bits=%d", n_bits);
  TQ_REQUIRE(x != nullptr && n > 0, "tq_calibrate_minmax: NULL / empty input");
  TQ_REQUIRE(dtype == TQ_F32 || dtype == TQ_BF16 || dtype == TQ_F16, "tq_calibrate_minmax: bad dtype %d", dtype);
  if (y != nullptr && !symmetric && n_groups == 0 && n_params > 1 && inner > 1 && n % (n_params * inner) == 0) {
    // per-parameter data small enough for one block's registers: statistics, update, parameters and quantization in ONE
    // launch with one read of x (tq_fake_quant.hip, calib_rows_onepass_k); -1 = shape does not qualify
    const RowsOnePassArgs a{n / (n_params * inner), n_params, inner, mode, n_bits, log_domain, prev_min, prev_max, cur_min, cur_max,
                            delta, zero_float, eps, (float)(1.0 - momentum), (float)momentum};
    const int rc = launch_calib_rows_onepass(x, y, dtype, a, static_cast<hipStream_t>(stream));
    if (rc >= 0) return rc;
  }
  const size_t stats_bytes = (2 * n_params * sizeof(float) + 255) / 256 * 256;
  TQ_REQUIRE(workspace && workspace_bytes >= stats_bytes, "tq_calibrate_minmax: workspace too small");
  float* stats = static_cast<float*>(workspace);
  char* rest = static_cast<char*>(workspace) + stats_bytes;
  if (int e = tq_minmax(x, n, dtype, n_params, inner, stats, stats + n_params, rest, workspace_bytes - stats_bytes, stream))
    return e;
  const size_t lds = (2 * n_params + 2 * n_groups) * sizeof(float);
  hipLaunchKernelGGL(calib_update_k, dim3(1), dim3(n_params >= 256 ? 1024 : 256), lds, static_cast<hipStream_t>(stream), mode,
                     stats, stats + n_params, prev_min, prev_max, cur_min, cur_max, (uint32_t)n_params, momentum,
                     (uint32_t)n_groups, order, n_bits, symmetric, eps, log_domain, delta, zero_float, signed_flag, 0);
  if (int e = check_launch("calib_update_k")) return e;
  if (y != nullptr) {
    tq_quantizer q{delta, zero_float, signed_flag, n_bits, symmetric, log_domain, eps, n_params, inner};
    return tq_fake_quant_fwd(x, y, nullptr, TQ_IDX_NONE, n, dtype, &q, stream);
  }
  return TQ_OK;
}

constexpr unsigned kTicketMaxBlocks = 512;
constexpr unsigned kTicketFreeMaxBlocks = 2048;   // (TQ_CALIB_FREE_MAX overrides: A/B)   // partial pairs every quantizer block folds itself

// Statistics half of the ticket-free single-GPU step: per-block (min, max) -> ws[2 b], ws[2 b + 1]; block 0 also copies
// the previous estimator state behind the partials (ws[2 gridDim.x], ws[2 gridDim.x + 1]) so that the quantizer launch
// -- whose block 0 overwrites the state, possibly in place -- never reads the live buffers.
template <int DT, bool VEC>
__global__ __launch_bounds__(kBlock) void calib_partials_k(const void* __restrict__ x, uint64_t n, float* __restrict__ ws,
                                                           const float* prev_min, const float* prev_max) {
  constexpr int V = Store<DT>::kVec;
  typedef typename Store<DT>::elem_t E;
  __shared__ float s_red[2][kBlock / kWave];
  MinMax acc;
  if (VEC) {
    const u32x4* xv = static_cast<const u32x4*>(x);
    const uint64_t n_all = n / V;
    const uint64_t chunk = (n_all + gridDim.x - 1) / gridDim.x;
    const uint64_t n_vec = min(n_all, ((uint64_t)blockIdx.x + 1) * chunk);
    constexpr int U = 4;
    uint64_t i = (uint64_t)blockIdx.x * chunk + threadIdx.x;
    for (; i + (U - 1) * (uint64_t)kBlock < n_vec; i += U * (uint64_t)kBlock) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = ld_stream(xv + i + u * (uint64_t)kBlock);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float f[V];
        Store<DT>::unpack(v[u], f);
#pragma unroll
        for (int j = 0; j < V; ++j) acc.add(f[j]);
      }
    }
    for (; i < n_vec; i += kBlock) {
      float f[V];
      Store<DT>::unpack(xv[i], f);
#pragma unroll
      for (int j = 0; j < V; ++j) acc.add(f[j]);
    }
    if (blockIdx.x == 0 && n_all * V + threadIdx.x < n)                      // ragged tail (< V elements)
      acc.add(Store<DT>::load1(static_cast<const E*>(x) + n_all * V + threadIdx.x));
  } else {
    const E* xs = static_cast<const E*>(x);
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock)
      acc.add(Store<DT>::load1(xs + i));
  }
  const int w = threadIdx.x / kWave;
  float mn = wave_min(acc.lo()), mx = wave_max(acc.hi());
  if ((threadIdx.x & (kWave - 1)) == 0) { s_red[0][w] = mn; s_red[1][w] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < kBlock / kWave; ++k) { mn = min_nanprop(mn, s_red[0][k]); mx = max_nanprop(mx, s_red[1][k]); }
    ws[2 * blockIdx.x] = mn;
    ws[2 * blockIdx.x + 1] = mx;
    if (blockIdx.x == 0 && prev_min != nullptr) {
      ws[2 * gridDim.x] = prev_min[0];
      ws[2 * gridDim.x + 1] = prev_max[0];
    }
  }
}

extern "C" int tq_calibrate_tensor(const void* x, uint64_t n, int dtype, int mode, const float* prev_min,
                                   const float* prev_max, float* cur_min, float* cur_max, double momentum, int n_bits,
                                   int symmetric, float eps, int log_domain, float* delta, float* zero_float,
                                   uint8_t* signed_flag, void* y, void* workspace, size_t workspace_bytes,
                                   uint32_t* counter, tq_stream_t stream) {
  TQ_REQUIRE(x && n > 0, "tq_calibrate_tensor: empty tensor has no min/max");
  TQ_REQUIRE(cur_min && cur_max && delta && counter, "tq_calibrate_tensor: NULL output");
  TQ_REQUIRE((prev_min == nullptr) == (prev_max == nullptr), "tq_calibrate_tensor: prev_min / prev_max mismatch");
  TQ_REQUIRE(symmetric ? signed_flag != nullptr : zero_float != nullptr, "tq_calibrate_tensor: missing parameter output");
  TQ_REQUIRE(mode >= TQ_EST_CURRENT && mode <= TQ_EST_RUNNING, "tq_calibrate_tensor: bad mode %d", mode);
  TQ_REQUIRE(n_bits >= 1 && n_bits <= 24, "tq_calibrate_tensor: n_bits=%d", n_bits);
  TQ_REQUIRE(dtype == TQ_F32 || dtype == TQ_BF16 || dtype == TQ_F16, "tq_calibrate_tensor: bad dtype %d", dtype);
  const int V = dtype == TQ_F32 ? 4 : 8;
  const MMPlan pl = plan_minmax(n, 1, 1, V, aligned16(x));
  // Every block pays a device-scope release (buffer_wbl2 + buffer_inv: the 8 XCD L2s are not coherent with each
  // other) before it takes its ticket: ~3.5 us x blocks / resident blocks.  Worth it while launches dominate
  // (small tensors); a [1024,512,768] tensor has 16384 blocks and ran 930 us against 125 us for the separate
  // statistics + finalize + update launches, so large tensors take that path.
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* ws = static_cast<float*>(workspace);
  const bool vec = aligned16(x);
  // ticket-free form when the quantized tensor is wanted too (the usual calibrating forward): block partials + copy of
  // the previous state in launch 1, fold + update + parameters + quantize in launch 2 (fq_tensor_calib).  Every block of
  // launch 2 folds all partial pairs itself (8 bytes each, from L2): taken up to kTicketFreeMaxBlocks partials (round 6:
  // was 512, so that a [16384,768] site of a [128,128] calibrating forward ran statistics -> finalize -> update -> quantize
  // as four launches, ~9.5 us of which are the two parameter-sized ones: 159 sites per BERT-base forward).
  static const int ticket_free = tuning("TQ_CALIB_TICKET_FREE", 1);
  static const unsigned free_max = (unsigned)tuning("TQ_CALIB_FREE_MAX", (int)kTicketFreeMaxBlocks);
  const bool free_ok = ticket_free && y != nullptr && vec && aligned16(y) && pl.gx <= free_max && workspace &&
                       workspace_bytes >= ((size_t)pl.gx * 2 + 2) * sizeof(float);
  if (pl.gx > kTicketMaxBlocks && !free_ok)
    return tq_calibrate_minmax(x, n, dtype, 1, 1, mode, prev_min, prev_max, cur_min, cur_max, momentum, 0, nullptr, n_bits,
                               symmetric, eps, log_domain, delta, zero_float, signed_flag, y, workspace, workspace_bytes, stream);
  TQ_REQUIRE(workspace && workspace_bytes >= (size_t)pl.gx * 2 * sizeof(float), "tq_calibrate_tensor: workspace too small");
  if (free_ok) {
    switch (dtype) {
      case TQ_F32: hipLaunchKernelGGL((calib_partials_k<TQ_F32, true>), dim3(pl.gx), dim3(kBlock), 0, st, x, n, ws, prev_min, prev_max); break;
      case TQ_BF16: hipLaunchKernelGGL((calib_partials_k<TQ_BF16, true>), dim3(pl.gx), dim3(kBlock), 0, st, x, n, ws, prev_min, prev_max); break;
      default: hipLaunchKernelGGL((calib_partials_k<TQ_F16, true>), dim3(pl.gx), dim3(kBlock), 0, st, x, n, ws, prev_min, prev_max); break;
    }
    if (int e = check_launch("calib_partials_k")) return e;
    const float* pm = prev_min ? ws + 2 * pl.gx : nullptr;
    const float* px = prev_min ? ws + 2 * pl.gx + 1 : nullptr;
    CalibApplyArgs c{nullptr, ws, (uint32_t)pl.gx, pm, px, cur_min, cur_max, delta, zero_float, signed_flag, mode, n_bits,
                     symmetric, log_domain, eps, (float)(1.0 - momentum), (float)momentum};
    return launch_fq_from_stats(x, y, n, dtype, c, st);
  }
  float* stats_out = nullptr;
#define TQ_CALIB(DTV)                                                                                              \
  if (vec) hipLaunchKernelGGL((calib_tensor_k<DTV, true>), dim3(pl.gx), dim3(kBlock), 0, st, x, n, ws, counter, mode, prev_min, \
                              prev_max, cur_min, cur_max, momentum, n_bits, symmetric, eps, log_domain, delta, zero_float,     \
                              signed_flag, stats_out);                                                               \
  else hipLaunchKernelGGL((calib_tensor_k<DTV, false>), dim3(pl.gx), dim3(kBlock), 0, st, x, n, ws, counter, mode, prev_min,   \
                          prev_max, cur_min, cur_max, momentum, n_bits, symmetric, eps, log_domain, delta, zero_float,         \
                          signed_flag, stats_out)
  switch (dtype) {
    case TQ_F32: TQ_CALIB(TQ_F32); break;
    case TQ_BF16: TQ_CALIB(TQ_BF16); break;
    default: TQ_CALIB(TQ_F16); break;
  }
  if (int e = check_launch("calib_tensor_k")) return e;
  if (y != nullptr) {
    tq_quantizer q{delta, zero_float, signed_flag, n_bits, symmetric, log_domain, eps, 1, 1};
    return tq_fake_quant_fwd(x, y, nullptr, TQ_IDX_NONE, n, dtype, &q, stream);
  }
  return TQ_OK;
}

// ---- sharded calibration: the fused step split at the exchange ---------------------------------------------
// stats: fp32 [2 * n_params] = [-min | max] of the LOCAL shard, written by the statistics kernel itself; the
// caller all-reduces it in place with MAX (one collective: min and max fused) and hands it to tq_calibrate_apply.
// copy_prev_*: non-NULL (one range, ticket path only) -> stats[2], stats[3] receive the previous state; returns 1 in
// *copied when that happened
static int calibrate_stats_impl(const void* x, uint64_t n, int dtype, uint64_t n_params, uint64_t inner, float* stats,
                                void* workspace, size_t workspace_bytes, uint32_t* counter, tq_stream_t stream,
                                const float* copy_prev_min, const float* copy_prev_max, int* copied) {
  if (copied) *copied = 0;
  TQ_REQUIRE(x && n > 0 && stats, "tq_calibrate_stats: empty tensor / NULL output");
  TQ_REQUIRE(n_params >= 1 && n_params <= kCalibMaxN, "tq_calibrate_stats: n_params=%llu > %u", (unsigned long long)n_params, kCalibMaxN);
  TQ_REQUIRE(dtype == TQ_F32 || dtype == TQ_BF16 || dtype == TQ_F16, "tq_calibrate_stats: bad dtype %d", dtype);
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* ws = static_cast<float*>(workspace);
  if (n_params == 1 && counter != nullptr) {
    const int V = dtype == TQ_F32 ? 4 : 8;
    const MMPlan pl = plan_minmax(n, 1, 1, V, aligned16(x));
    if (pl.gx <= kTicketMaxBlocks) {      // one launch: the last-ticket block writes [-min, max]
      TQ_REQUIRE(workspace && workspace_bytes >= (size_t)pl.gx * 2 * sizeof(float), "tq_calibrate_stats: workspace too small");
      const bool vec = aligned16(x);
      float* stats_out = stats;
      const int mode = TQ_EST_CURRENT, n_bits = 8, symmetric = 0, log_domain = 0;
      const float *prev_min = copy_prev_min, *prev_max = copy_prev_max;     // (stats-only mode: copied, not applied)
      if (copied) *copied = copy_prev_min != nullptr;
      float *cur_min = nullptr, *cur_max = nullptr, *delta = nullptr, *zero_float = nullptr;
      uint8_t* signed_flag = nullptr;
      const double momentum = 0.0;
      const float eps = 0.0f;
      switch (dtype) {
        case TQ_F32: TQ_CALIB(TQ_F32); break;
        case TQ_BF16: TQ_CALIB(TQ_BF16); break;
        default: TQ_CALIB(TQ_F16); break;
      }
      return check_launch("calib_tensor_k(stats)");
    }
  }
  if (n_params > 1) {
    TQ_REQUIRE(inner >= 1 && n % (n_params * inner) == 0, "tq_calibrate_stats: n not a multiple of n_params*inner");
  }
  switch (dtype) {
    case TQ_F32: return launch_minmax<TQ_F32>(x, n, n_params, inner, stats, stats + n_params, ws, workspace_bytes, st, true);
    case TQ_BF16: return launch_minmax<TQ_BF16>(x, n, n_params, inner, stats, stats + n_params, ws, workspace_bytes, st, true);
    default: return launch_minmax<TQ_F16>(x, n, n_params, inner, stats, stats + n_params, ws, workspace_bytes, st, true);
  }
}
#undef TQ_CALIB

extern "C" int tq_calibrate_stats(const void* x, uint64_t n, int dtype, uint64_t n_params, uint64_t inner, float* stats,
                                  void* workspace, size_t workspace_bytes, uint32_t* counter, tq_stream_t stream) {
  return calibrate_stats_impl(x, n, dtype, n_params, inner, stats, workspace, workspace_bytes, counter, stream, nullptr,
                              nullptr, nullptr);
}

// The one-call sharded steps (tq_calibrate_minmax_rccl / _mailbox): statistics into a scratch `stats` of >= 4 floats,
// with the previous state copied behind them when the single-range ticket path runs; *prev_in_stats tells the second
// half (calibrate_apply_after_exchange) to read that copy -- the fused update + quantize launch is then safe for
// in-place state, i.e. inside a captured hipGraph.
int tq::calibrate_stats_for_exchange(const void* x, uint64_t n, int dtype, uint64_t n_params, uint64_t inner, float* stats,
                                     void* workspace, size_t workspace_bytes, uint32_t* counter, const float* prev_min,
                                     const float* prev_max, int* prev_in_stats, tq_stream_t stream) {
  const bool one = n_params == 1 && prev_min != nullptr;
  return calibrate_stats_impl(x, n, dtype, n_params, inner, stats, workspace, workspace_bytes, counter, stream,
                              one ? prev_min : nullptr, one ? prev_max : nullptr, prev_in_stats);
}

// stats: [-min | max] over ALL ranks -> estimator update + range -> parameters (one launch) [-> quantize x].
static int calibrate_apply_impl(const float* stats, const void* x, uint64_t n, int dtype, uint64_t n_params,
                                uint64_t inner, int mode, const float* prev_min, const float* prev_max, float* cur_min,
                                float* cur_max, double momentum, uint64_t n_groups, const int64_t* order, int n_bits,
                                int symmetric, float eps, int log_domain, float* delta, float* zero_float,
                                uint8_t* signed_flag, void* y, tq_stream_t stream, int prev_in_stats);

int tq::calibrate_apply_after_exchange(const float* stats, const void* x, uint64_t n, int dtype, uint64_t n_params,
                                       uint64_t inner, int mode, const float* prev_min, const float* prev_max,
                                       float* cur_min, float* cur_max, double momentum, uint64_t n_groups,
                                       const int64_t* order, int n_bits, int symmetric, float eps, int log_domain,
                                       float* delta, float* zero_float, uint8_t* signed_flag, void* y, int prev_in_stats,
                                       tq_stream_t stream) {
  return calibrate_apply_impl(stats, x, n, dtype, n_params, inner, mode, prev_min, prev_max, cur_min, cur_max, momentum,
                              n_groups, order, n_bits, symmetric, eps, log_domain, delta, zero_float, signed_flag, y, stream,
                              prev_in_stats);
}

extern "C" int tq_calibrate_apply(const float* stats, const void* x, uint64_t n, int dtype, uint64_t n_params,
                                  uint64_t inner, int mode, const float* prev_min, const float* prev_max, float* cur_min,
                                  float* cur_max, double momentum, uint64_t n_groups, const int64_t* order, int n_bits,
                                  int symmetric, float eps, int log_domain, float* delta, float* zero_float,
                                  uint8_t* signed_flag, void* y, tq_stream_t stream) {
  return calibrate_apply_impl(stats, x, n, dtype, n_params, inner, mode, prev_min, prev_max, cur_min, cur_max, momentum,
                              n_groups, order, n_bits, symmetric, eps, log_domain, delta, zero_float, signed_flag, y, stream, 0);
}

static int calibrate_apply_impl(const float* stats, const void* x, uint64_t n, int dtype, uint64_t n_params,
                                uint64_t inner, int mode, const float* prev_min, const float* prev_max, float* cur_min,
                                float* cur_max, double momentum, uint64_t n_groups, const int64_t* order, int n_bits,
                                int symmetric, float eps, int log_domain, float* delta, float* zero_float,
                                uint8_t* signed_flag, void* y, tq_stream_t stream, int prev_in_stats) {
  TQ_REQUIRE(stats && cur_min && cur_max && delta, "tq_calibrate_apply: NULL pointer");
  TQ_REQUIRE((prev_min == nullptr) == (prev_max == nullptr), "tq_calibrate_apply: prev_min / prev_max mismatch");
  TQ_REQUIRE(symmetric ? signed_flag != nullptr : zero_float != nullptr, "tq_calibrate_apply: missing parameter output");
  TQ_REQUIRE(mode >= TQ_EST_CURRENT && mode <= TQ_EST_RUNNING, "tq_calibrate_apply: bad mode %d", mode);
  TQ_REQUIRE(n_params >= 1 && n_params <= kCalibMaxN, "tq_calibrate_apply: n_params=%llu > %u", (unsigned long long)n_params, kCalibMaxN);
  TQ_REQUIRE(n_groups == 0 || n_params % n_groups == 0, "tq_calibrate_apply: n_params %% n_groups != 0");
  TQ_REQUIRE(n_bits >= 1 && n_bits <= 24, "tq_calibrate_apply: n_bits=%d", n_bits);
  // one range, fresh output buffers, vectorisable tensor: update + parameters + quantize as ONE launch (every block
  // re-derives the dozen scalars; in-place state -- the hipGraph mode -- keeps the separate update launch below, since
  // block 0's store could race another block's read of the previous state)
  static const int fused_apply = tuning("TQ_CALIB_FUSED_APPLY", 1);
  if (prev_in_stats) {                  // the statistics launch left a copy of the previous state behind the statistics
    prev_min = stats + 2;
    prev_max = stats + 3;
  }
  if (fused_apply && n_params == 1 && n_groups == 0 && y != nullptr && x != nullptr && aligned16(x) && aligned16(y) &&
      (prev_in_stats || (cur_min != prev_min && cur_max != prev_max)) &&
      (dtype == TQ_F32 || dtype == TQ_BF16 || dtype == TQ_F16)) {
    CalibApplyArgs c{stats, nullptr, 0u, prev_min, prev_max, cur_min, cur_max, delta, zero_float, signed_flag, mode, n_bits,
                     symmetric, log_domain, eps, (float)(1.0 - momentum), (float)momentum};
    return launch_fq_from_stats(x, y, n, dtype, c, static_cast<hipStream_t>(stream));
  }
  const size_t lds = (2 * n_params + 2 * n_groups) * sizeof(float);
  hipLaunchKernelGGL(calib_update_k, dim3(1), dim3(n_params >= 256 ? 1024 : 256), lds, static_cast<hipStream_t>(stream), mode,
                     stats, stats + n_params, prev_min, prev_max, cur_min, cur_max, (uint32_t)n_params, momentum,
                     (uint32_t)n_groups, order, n_bits, symmetric, eps, log_domain, delta, zero_float, signed_flag, 1);
  if (int e = check_launch("calib_update_k")) return e;
  if (y != nullptr) {
    TQ_REQUIRE(x != nullptr, "tq_calibrate_apply: y without x");
    tq_quantizer q{delta, zero_float, signed_flag, n_bits, symmetric, log_domain, eps, n_params, inner};
    return tq_fake_quant_fwd(x, y, nullptr, TQ_IDX_NONE, n, dtype, &q, stream);
  }
  return TQ_OK;
}
