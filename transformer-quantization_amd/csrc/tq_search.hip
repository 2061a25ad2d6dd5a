// K7/K8/K9: range-search losses for many candidate quantizers in ONE pass over the tensor.
//
// The reference evaluates candidates one at a time: deepcopy(quantizer) + set_quant_range +
// fake-quant + (x-y)^2 + sum, ~9 tensor sweeps and a host sync per candidate
// (range_estimators.py:248-256, 287-294, 356-420): 100 candidates for the 1-D grid,
// 100 x 64 x 2 for the asymmetric 8-bit 2-D grid.  Here a block keeps a tile of x in registers
// (16 values per lane) and walks a tile of candidates whose (scale, zp, lo, hi) sit in LDS and
// are read as wave-uniform broadcasts; x is read from HBM once per candidate tile (later tiles
// hit L2 / MALL).  VALU-bound by design: ~25 fp32 ops per element per candidate, of which the
// IEEE division is ~11.
//
// Accumulation: fp32 over the 16 register values, fp64 across lanes / tiles / blocks.  The
// reference sums in fp32 (torch.sum) and accumulates batches in fp64 (numpy); parity is
// "same argmin, or equal loss within 1e-6 relative" (SURVEY.md section 7).
#include <algorithm>

#include "tq_device.h"
#include "tq_host.h"

namespace tq {

constexpr int kCandTile = 128;

// A "row" (one loss vector) is either contiguous (seg == row_len) or the union of segments of `seg`
// elements that repeat every `seg_stride` elements: row r, element e lives at
//   (e / seg) * seg_stride + r * seg + (e % seg)
// which is how a group of embedding dimensions [r*seg, (r+1)*seg) of a [tokens, d] tensor is laid out
// (seg = d / n_groups, seg_stride = d): per-group searches need no transpose copy.
template <int DT, int E>
__global__ __launch_bounds__(kBlock) void mse_cand_k(const void* __restrict__ x, uint64_t row_len, uint64_t seg,
                                                     uint64_t seg_stride, bool vec_ok,
                                                     const float4* __restrict__ cand, uint32_t n_cand,
                                                     uint32_t cand_tile, double* __restrict__ partial) {
  constexpr int V = Store<DT>::kVec;
  constexpr int NV = E / V;   // 16-byte vectors per lane per tile
  typedef typename Store<DT>::elem_t T;
  __shared__ float4 s_c[kCandTile];
  __shared__ float s_r[kCandTile];                 // guarded reciprocal of each candidate's scale
  __shared__ double s_acc[kBlock / kWave][kCandTile];

  const uint32_t c0 = blockIdx.y * cand_tile;
  const uint32_t nc = min(cand_tile, n_cand - c0);
  for (uint32_t c = threadIdx.x; c < kCandTile; c += kBlock) {
    if (c < nc) {
      const float4 pc = cand[c0 + c];
      s_c[c] = pc;
      s_r[c] = guarded_rcp(pc.x);
    }
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) s_acc[w][c] = 0.0;
  }
  __syncthreads();

  const uint64_t row = blockIdx.z;
  const bool contiguous = seg == row_len;
  const T* xr = static_cast<const T*>(x) + row * seg;
  const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
  constexpr uint64_t kTile = (uint64_t)kBlock * E;

  for (uint64_t t0 = (uint64_t)blockIdx.x * kTile; t0 < row_len; t0 += (uint64_t)gridDim.x * kTile) {
    float f[E];
    if (vec_ok && contiguous && t0 + kTile <= row_len) {
      const u32x4* xv = reinterpret_cast<const u32x4*>(xr + t0);
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        float g[V];
        Store<DT>::unpack(xv[u * kBlock + threadIdx.x], g);
#pragma unroll
        for (int j = 0; j < V; ++j) f[u * V + j] = g[j];
      }
    } else if (vec_ok && !contiguous) {
      // segmented row: seg % V == 0, so a 16-byte vector never straddles two segments
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const uint64_t e = t0 + ((uint64_t)u * kBlock + threadIdx.x) * V;
        float g[V];
#pragma unroll
        for (int j = 0; j < V; ++j) g[j] = 0.0f;
        if (e < row_len) {
          const uint64_t sidx = e / seg, off = e - sidx * seg;
          Store<DT>::unpack(*reinterpret_cast<const u32x4*>(xr + sidx * seg_stride + off), g);
        }
#pragma unroll
        for (int j = 0; j < V; ++j) f[u * V + j] = g[j];
      }
    } else if (!contiguous) {
#pragma unroll
      for (int j = 0; j < E; ++j) {
        const uint64_t e = t0 + (uint64_t)j * kBlock + threadIdx.x;
        const uint64_t sidx = e / seg, off = e - sidx * seg;
        f[j] = e < row_len ? Store<DT>::load1(xr + sidx * seg_stride + off) : 0.0f;
      }
    } else {
      // ragged / unaligned tile: zero padding contributes exactly 0 to every candidate's loss
      // (0 quantizes to 0: round(0/s)+zp = zp lies inside [lo, hi]).
#pragma unroll
      for (int j = 0; j < E; ++j) {
        const uint64_t k = t0 + (uint64_t)j * kBlock + threadIdx.x;
        f[j] = k < row_len ? Store<DT>::load1(xr + k) : 0.0f;
      }
    }
    for (uint32_t c = 0; c < nc; ++c) {
      const float4 pc = s_c[c];
      const QP p = {pc.x, pc.y, pc.z, pc.w};
      // Two elements per instruction where the ISA has packed fp32 forms (v_pk_mul / v_pk_add / v_pk_fma_f32):
      // this kernel is VALU-bound (~14 fp32 ops per element and candidate), and rounding / compares / selects
      // stay scalar.  Per-element semantics are those of rne_quot1 + clamp + dequant (tq_device.h).
      const float rc = s_r[c];
      const f32x2 r2 = {rc, rc}, s2 = {p.scale, p.scale}, z2 = {p.zp, p.zp};
      const f32x2 tol2 = {-kTieTol, -kTieTol}, half2 = {0.5f - kTieTol, 0.5f - kTieTol};
      f32x2 acc2 = {0.0f, 0.0f};
#pragma unroll
      for (int j = 0; j < E; j += 2) {
        const f32x2 x = {f[j], f[j + 1]};
        const f32x2 q0 = x * r2;
        f32x2 h = {rintf(q0.x), rintf(q0.y)};
        const f32x2 thr = __builtin_elementwise_fma(__builtin_elementwise_abs(h), tol2, half2);
        const f32x2 off = q0 - h;
        if (!(fabsf(off.x) < thr.x)) h.x = rintf(x.x / p.scale);      // doubtful elements: true division
        if (!(fabsf(off.y) < thr.y)) h.y = rintf(x.y / p.scale);
        f32x2 xi = h + z2;
        // v_med3_f32 (1 op) instead of the NaN-propagating clamp (4 ops): a NaN input still poisons this
        // candidate's loss through x - dequant below, +-Inf clamps to the grid ends either way
        xi.x = __builtin_amdgcn_fmed3f(xi.x, p.lo, p.hi);
        xi.y = __builtin_amdgcn_fmed3f(xi.y, p.lo, p.hi);
        const f32x2 d = x - s2 * (xi - z2);
        acc2 = acc2 + d * d;
      }
      const float acc = acc2.x + acc2.y;
      const double tot = wave_sum((double)acc);
      if (lane == 0) s_acc[wave][c] += tot;
    }
  }
  __syncthreads();
  for (uint32_t c = threadIdx.x; c < nc; c += kBlock) {
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) tot += s_acc[w][c];
    partial[(row * gridDim.x + blockIdx.x) * n_cand + c0 + c] = tot;
  }
}

// loss[row][c] += sum_b partial[row][b][c]
__global__ void mse_final_k(const double* __restrict__ partial, uint32_t nb, uint32_t n_cand, double* __restrict__ loss) {
  const uint64_t row = blockIdx.y;
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cand) return;
  double tot = 0.0;
  for (uint32_t b = 0; b < nb; ++b) tot += partial[(row * nb + b) * n_cand + c];
  loss[row * n_cand + c] += tot;
}

struct MsePlan { unsigned gx, gy; int e; unsigned cand_tile; };

static MsePlan plan_mse(uint64_t rows, uint64_t row_len, uint64_t n_cand) {
  MsePlan p;
  p.e = row_len <= 2048 ? 4 : 16;
  if (row_len > 2048 && row_len < (uint64_t)kBlock * 16 * 256) p.e = 8;   // more, smaller tiles for mid sizes
  const uint64_t tile = (uint64_t)kBlock * p.e;
  const uint64_t tiles = ceil_div(row_len, tile);
  // small tensors: split the candidates over more blocks (x is re-read from L2) until the chip is full
  p.cand_tile = kCandTile;
  while (p.cand_tile > 16 && tiles * rows * ceil_div(n_cand, p.cand_tile) < 2048) p.cand_tile /= 2;
  p.gy = (unsigned)ceil_div(n_cand, p.cand_tile);
  const uint64_t budget = std::max<uint64_t>(1, (uint64_t)4096 / std::max<uint64_t>(1, (uint64_t)p.gy * rows));
  p.gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(tiles, budget));
  return p;
}

// ---------------------------------------------------------------------------------- xent
// One thread per candidate; x is a tiny [rows, cols] logits matrix (cols <= 64).
__global__ void xent_cand_k(const float* __restrict__ x, uint32_t rows, uint32_t cols, const float4* __restrict__ cand,
                            uint32_t n_cand, double* __restrict__ loss) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cand) return;
  const float4 pc = cand[c];
  const QP p = {pc.x, pc.y, pc.z, pc.w};
  double tot = 0.0;
  for (uint32_t r = 0; r < rows; ++r) {
    const float* xr = x + (uint64_t)r * cols;
    float xm = -__builtin_huge_valf(), qm = -__builtin_huge_valf();
    for (uint32_t j = 0; j < cols; ++j) {
      xm = fmaxf(xm, xr[j]);
      qm = fmaxf(qm, q_dequant(q_index(xr[j], p), p));
    }
    float xs = 0.f, qs = 0.f;
    for (uint32_t j = 0; j < cols; ++j) {
      xs += expf(xr[j] - xm);
      qs += expf(q_dequant(q_index(xr[j], p), p) - qm);
    }
    const float lqs = logf(qs);
    float row_loss = 0.f;
    for (uint32_t j = 0; j < cols; ++j) {
      const float prob = expf(xr[j] - xm) / xs;                                   // softmax(x)
      const float logq = (q_dequant(q_index(xr[j], p), p) - qm) - lqs;            // log_softmax(Q(x))
      row_loss += -prob * logq;
    }
    tot += (double)row_loss;
  }
  loss[c] += tot;
}

// ---------------------------------------------------------------------------------- argmin
// np.argmin semantics: first minimum; a NaN counts as the minimum (first NaN wins).
__global__ void argmin_select_k(const double* __restrict__ loss, uint32_t n_cand, const float* __restrict__ thr_min,
                                const float* __restrict__ thr_max, float* __restrict__ cur_min,
                                float* __restrict__ cur_max, int64_t* __restrict__ best) {
  __shared__ double s_v[kBlock];
  __shared__ uint32_t s_i[kBlock];
  const uint64_t row = blockIdx.x;
  const double* l = loss + row * n_cand;
  double bv = __builtin_huge_val();
  uint32_t bi = 0xffffffffu;
  bool bnan = false;
  for (uint32_t c = threadIdx.x; c < n_cand; c += kBlock) {
    const double v = l[c];
    const bool vnan = v != v;
    const bool better = bi == 0xffffffffu || (vnan && !bnan) || (!bnan && !vnan && v < bv);
    if (better) { bv = v; bi = c; bnan = vnan; }
  }
  s_v[threadIdx.x] = bv;
  s_i[threadIdx.x] = bi;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < kBlock; ++k) {
      const uint32_t oi = s_i[k];
      if (oi == 0xffffffffu) continue;
      const double ov = s_v[k];
      const bool onan = ov != ov;
      bool better;
      if (bi == 0xffffffffu) better = true;
      else if (onan != bnan) better = onan;
      else if (onan) better = oi < bi;
      else better = ov < bv || (ov == bv && oi < bi);
      if (better) { bv = ov; bi = oi; bnan = onan; }
    }
    if (best) best[row] = (int64_t)bi;
    cur_min[row] = thr_min[bi];
    cur_max[row] = thr_max[bi];
  }
}

template <int DT>
static int launch_mse(const void* x, uint64_t rows, uint64_t row_len, uint64_t seg, uint64_t seg_stride,
                      const float* cand, uint64_t n_cand, double* loss, double* ws, size_t ws_bytes, hipStream_t st) {
  constexpr int V = Store<DT>::kVec;
  const MsePlan pl = plan_mse(rows, row_len, n_cand);
  const size_t need = (size_t)rows * pl.gx * n_cand * sizeof(double);
  if (ws == nullptr || ws_bytes < need) return set_error(TQ_EWORKSPACE, "tq_mse_candidates: workspace %zu < %zu", ws_bytes, need);
  const bool contiguous = seg == row_len;
  const bool vec_ok = contiguous
      ? aligned16(x) && ((row_len * elem_size(DT)) % 16 == 0 || rows == 1) && (row_len % V == 0 || rows == 1)
      : aligned16(x) && seg % V == 0 && seg_stride % V == 0;
  const dim3 grid(pl.gx, pl.gy, (unsigned)rows);
  const float4* c4 = reinterpret_cast<const float4*>(cand);
  if (pl.e == 16) hipLaunchKernelGGL((mse_cand_k<DT, 16>), grid, dim3(kBlock), 0, st, x, row_len, seg, seg_stride, vec_ok, c4, (uint32_t)n_cand, pl.cand_tile, ws);
  else if (pl.e == 8) hipLaunchKernelGGL((mse_cand_k<DT, 8>), grid, dim3(kBlock), 0, st, x, row_len, seg, seg_stride, vec_ok, c4, (uint32_t)n_cand, pl.cand_tile, ws);
  else hipLaunchKernelGGL((mse_cand_k<DT, 4>), grid, dim3(kBlock), 0, st, x, row_len, seg, seg_stride, vec_ok && V <= 4, c4, (uint32_t)n_cand, pl.cand_tile, ws);
  if (int e = check_launch("mse_cand_k")) return e;
  hipLaunchKernelGGL(mse_final_k, dim3((unsigned)ceil_div(n_cand, 256), (unsigned)rows), dim3(256), 0, st, ws, pl.gx,
                     (uint32_t)n_cand, loss);
  return check_launch("mse_final_k");
}

}  // namespace tq

using namespace tq;

extern "C" size_t tq_mse_workspace_bytes(uint64_t rows, uint64_t row_len, uint64_t n_cand) {
  if (rows == 0 || row_len == 0 || n_cand == 0) return 0;
  const MsePlan pl = plan_mse(rows, row_len, n_cand);
  return (size_t)rows * pl.gx * n_cand * sizeof(double);
}

extern "C" int tq_mse_candidates(const void* x, uint64_t rows, uint64_t row_len, int dtype, const float* cand,
                                 uint64_t n_cand, double* loss, void* workspace, size_t workspace_bytes,
                                 tq_stream_t stream) {
  TQ_REQUIRE(x && cand && loss, "tq_mse_candidates: NULL pointer");
  TQ_REQUIRE(rows >= 1 && rows <= 65535, "tq_mse_candidates: rows=%llu outside 1..65535", (unsigned long long)rows);
  TQ_REQUIRE(n_cand >= 1 && n_cand < (1ull << 31), "tq_mse_candidates: bad candidate count");
  TQ_REQUIRE(dtype == TQ_F32 || dtype == TQ_BF16 || dtype == TQ_F16, "tq_mse_candidates: bad dtype %d", dtype);
  TQ_REQUIRE((reinterpret_cast<uintptr_t>(cand) & 15u) == 0, "tq_mse_candidates: candidate table must be 16-byte aligned");
  if (row_len == 0) return TQ_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* ws = static_cast<double*>(workspace);
  switch (dtype) {
    case TQ_F32: return launch_mse<TQ_F32>(x, rows, row_len, row_len, row_len, cand, n_cand, loss, ws, workspace_bytes, st);
    case TQ_BF16: return launch_mse<TQ_BF16>(x, rows, row_len, row_len, row_len, cand, n_cand, loss, ws, workspace_bytes, st);
    default: return launch_mse<TQ_F16>(x, rows, row_len, row_len, row_len, cand, n_cand, loss, ws, workspace_bytes, st);
  }
}

extern "C" int tq_mse_candidates_grouped(const void* x, uint64_t n_tokens, uint64_t d, uint64_t n_groups, int dtype,
                                         const float* cand, uint64_t n_cand, double* loss, void* workspace,
                                         size_t workspace_bytes, tq_stream_t stream) {
  TQ_REQUIRE(x && cand && loss, "tq_mse_candidates_grouped: NULL pointer");
  TQ_REQUIRE(n_groups >= 1 && n_groups <= 65535 && d % n_groups == 0, "tq_mse_candidates_grouped: d %% n_groups != 0");
  TQ_REQUIRE(n_cand >= 1 && n_cand < (1ull << 31), "tq_mse_candidates_grouped: bad candidate count");
  TQ_REQUIRE(dtype == TQ_F32 || dtype == TQ_BF16 || dtype == TQ_F16, "tq_mse_candidates_grouped: bad dtype %d", dtype);
  TQ_REQUIRE((reinterpret_cast<uintptr_t>(cand) & 15u) == 0, "tq_mse_candidates_grouped: candidate table must be 16-byte aligned");
  if (n_tokens == 0) return TQ_OK;
  const uint64_t gs = d / n_groups;
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* ws = static_cast<double*>(workspace);
  switch (dtype) {
    case TQ_F32: return launch_mse<TQ_F32>(x, n_groups, n_tokens * gs, gs, d, cand, n_cand, loss, ws, workspace_bytes, st);
    case TQ_BF16: return launch_mse<TQ_BF16>(x, n_groups, n_tokens * gs, gs, d, cand, n_cand, loss, ws, workspace_bytes, st);
    default: return launch_mse<TQ_F16>(x, n_groups, n_tokens * gs, gs, d, cand, n_cand, loss, ws, workspace_bytes, st);
  }
}

extern "C" int tq_xent_candidates(const float* x, uint64_t rows, uint64_t cols, const float* cand, uint64_t n_cand,
                                  double* loss, tq_stream_t stream) {
  TQ_REQUIRE(x && cand && loss, "tq_xent_candidates: NULL pointer");
  TQ_REQUIRE(rows >= 1 && cols >= 1 && n_cand >= 1, "tq_xent_candidates: empty input");
  TQ_REQUIRE((reinterpret_cast<uintptr_t>(cand) & 15u) == 0, "tq_xent_candidates: candidate table must be 16-byte aligned");
  hipLaunchKernelGGL(xent_cand_k, dim3((unsigned)ceil_div(n_cand, 64)), dim3(64), 0, static_cast<hipStream_t>(stream), x,
                     (uint32_t)rows, (uint32_t)cols, reinterpret_cast<const float4*>(cand), (uint32_t)n_cand, loss);
  return check_launch("xent_cand_k");
}

extern "C" int tq_argmin_select(const double* loss, uint64_t rows, uint64_t n_cand, const float* thr_min,
                                const float* thr_max, float* cur_min, float* cur_max, int64_t* best,
                                tq_stream_t stream) {
  TQ_REQUIRE(loss && thr_min && thr_max && cur_min && cur_max, "tq_argmin_select: NULL pointer");
  TQ_REQUIRE(rows >= 1 && n_cand >= 1 && n_cand < 0xffffffffull, "tq_argmin_select: bad shape");
  hipLaunchKernelGGL(argmin_select_k, dim3((unsigned)rows), dim3(kBlock), 0, static_cast<hipStream_t>(stream), loss,
                     (uint32_t)n_cand, thr_min, thr_max, cur_min, cur_max, best);
  return check_launch("argmin_select_k");
}
