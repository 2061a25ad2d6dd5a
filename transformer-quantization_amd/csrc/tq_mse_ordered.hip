// K7/K8 in the reference's own summation order.
//
// MSE_Estimator.loss_fx (reference quantization/range_estimators.py:248-256) returns
//     torch.sum( torch.sum(((data - Q(data)) ** 2).view(len(data), -1), dim=1) )
// as an fp32 value computed by ATen's CPU sum kernel (torch 2.10, aten/src/ATen/native/cpu/SumKernel.cpp,
// `cascade_sum`; third-party to the reference, restated in oracle/aten_sum.py and pinned there against
// torch.sum).  scipy's bounded Brent search (golden-section option, :296-327, :422-470) and the grid argmin
// consume that value, so reproducing the reference's thresholds bit for bit needs the same fp32 sum, not a
// more accurate one.  The order of a contiguous fp32 row of n >= 8 elements is:
//   * 32 independent accumulator columns (8-lane vectors x ilp_factor 4): column c adds elements
//     c, c+32, c+64, ... ("steps") sequentially into level 0 of a 4-level cascade; with
//     L = 2^max(4, ceil_log2(steps)/4), level j-1 is added into level j and cleared whenever the step index is
//     a multiple of L^j;
//   * at the end  ((l0 + l1) + l2) + l3  per column, then the 0..3 left-over 8-vectors are added to columns
//     0..7, then columns c, c+8, c+16, c+24 are folded in that order, then a scalar accumulator adds the
//     n % 8 tail elements followed by the 8 folded columns.
// Rows of fewer than 8 elements use the scalar variant (4 columns, no vectors).  The second torch.sum (over
// the len(data) row sums) is the same algorithm on that vector (single-threaded in ATen below 32768 rows).
//
// This maps onto a wavefront directly: a half-wave = the 32 columns, one lane per column, lane-local fp32
// accumulation across steps (no cross-lane traffic until the row ends).  Work unit = (row, span of L^k
// consecutive steps, tile of NC candidates); the candidate parameters are wave-uniform (SGPRs), x is read
// once per candidate tile (L2 hits after the first).  When a row is split into several units (few rows,
// few candidates) each unit emits its cascade state and `mse_ord_fold_k` replays the upper cascade levels
// in order; otherwise the unit finishes the row itself.  VALU-bound like the unordered kernel.
#include <algorithm>

#include "tq_device.h"
#include "tq_host.h"

namespace tq {

constexpr int kOrdNC = 8;            // candidates per unit (register accumulators: 4 levels x NC)

struct CandP {
  float scale, rcp, zp, lo, hi;
};

__device__ __forceinline__ CandP load_cand(const float4* __restrict__ cand, uint32_t c) {
  const float4 pc = cand[c];
  return {pc.x, guarded_rcp(pc.x), pc.y, pc.z, pc.w};
}

// (x - Q(x))^2 with the operation order of quantizers.py:184-185, 209 and range_estimators.py:250
// (each step rounded to fp32, no contraction).  v_med3 instead of the NaN-propagating clamp: a NaN input
// still poisons the loss through x - dequant.
__device__ __forceinline__ float sq_err(float x, const CandP& c) {
  const float h = rne_quot1(x, c.scale, c.rcp);
  const float xi = __builtin_amdgcn_fmed3f(h + c.zp, c.lo, c.hi);
  const float d = x - c.scale * (xi - c.zp);
  return d * d;
}

// element e of a contiguous row
template <int DT>
struct RowView {
  typedef typename Store<DT>::elem_t T;
  const T* base;
  __device__ __forceinline__ float operator()(uint64_t e) const { return Store<DT>::load1(base + e); }
};

struct PlainView {
  const float* base;
  uint64_t stride;
  __device__ __forceinline__ float operator()(uint64_t e) const { return base[e * stride]; }
};

// rows of fewer than 8 elements: ATen's scalar_inner_sum (row_sum with ilp_factor 4 on scalars)
template <class F>
__device__ __forceinline__ float ordered_sum_small(uint64_t n, F val) {
  float part[4] = {0.f, 0.f, 0.f, 0.f};
  const uint64_t size_ilp = n / 4;
  if (size_ilp) {
#pragma unroll
    for (int k = 0; k < 4; ++k) part[k] += val(k);
  }
  for (uint64_t i = size_ilp * 4; i < n; ++i) part[0] += val(i);
  part[0] += part[1];
  part[0] += part[2];
  part[0] += part[3];
  return part[0];
}

// End of a row for the half-wave that owns its 32 columns (col = lane & 31): cascade levels -> row sum.
template <class F>
__device__ __forceinline__ float ordered_finish(float l0, float l1, float l2, float l3, uint64_t n, uint64_t steps,
                                                uint32_t col, F val) {
  float pc = ((l0 + l1) + l2) + l3;
  const uint64_t vec = n / 8;
  for (uint64_t i = steps * 4; i < vec; ++i) pc += col < 8 ? val(i * 8 + col) : 0.0f;
  float p0 = pc;
  p0 += __shfl(pc, (int)col + 8, 32);
  p0 += __shfl(pc, (int)col + 16, 32);
  p0 += __shfl(pc, (int)col + 24, 32);
  float fin = 0.0f;
  for (uint64_t k = vec * 8; k < n; ++k) fin += val(k);
#pragma unroll
  for (int l = 0; l < 8; ++l) fin += __shfl(p0, l, 32);
  return fin;
}

// level j-1 -> level j dumps after a completed level-0 chunk ending at global step t (levels < k_top only)
#define TQ_ORD_DUMP(a, t, p, L, k_top, NCV)                                                  \
  do {                                                                                       \
    _Pragma("unroll") for (int c = 0; c < NCV; ++c) { a[1][c] += a[0][c]; a[0][c] = 0.0f; }  \
    if (k_top > 2 && (((t) >> (p)) & ((L) - 1)) == 0) {                                      \
      _Pragma("unroll") for (int c = 0; c < NCV; ++c) { a[2][c] += a[1][c]; a[1][c] = 0.0f; } \
      if (k_top > 3 && (((t) >> (2 * (p))) & ((L) - 1)) == 0) {                              \
        _Pragma("unroll") for (int c = 0; c < NCV; ++c) { a[3][c] += a[2][c]; a[2][c] = 0.0f; } \
      }                                                                                      \
    }                                                                                        \
  } while (0)

// Stage 1.  A wave = two half-waves working on consecutive (row, unit) items for the same candidate tile.
//   k_top = 4: one unit per row, the half-wave finishes the row -> row_loss[row, cand]
//   k_top = 2 / 3: unit = L^k_top steps, cascade state (levels 0..k_top-1) -> state[(row, unit), cand, col]
template <int DT, int NC>
__global__ __launch_bounds__(kBlock) void mse_ord_unit_k(const void* __restrict__ x, uint64_t row_len,
                                                         uint64_t steps, uint32_t p,
                                                         uint32_t k_top, uint64_t span, uint32_t units_per_row,
                                                         uint64_t n_ru, const float4* __restrict__ cand,
                                                         uint32_t n_cand, uint32_t n_ctiles,
                                                         float4* __restrict__ state, float* __restrict__ row_loss) {
  typedef typename Store<DT>::elem_t T;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t half = (threadIdx.x >> 5) & 1u, col = threadIdx.x & 31u;
  const uint64_t wid = (uint64_t)blockIdx.x * (kBlock / kWave) + wave;
  const uint32_t ctile = (uint32_t)(wid % n_ctiles);
  const uint64_t ru = (wid / n_ctiles) * 2 + half;
  if (ru >= n_ru) return;
  const uint64_t row = ru / units_per_row;
  const uint32_t u = (uint32_t)(ru - row * units_per_row);
  const uint64_t t0 = (uint64_t)u * span;
  const uint64_t t1 = min(steps, t0 + span);
  const uint32_t c0 = ctile * NC;
  const uint32_t L = 1u << p;

  CandP cp[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) cp[c] = load_cand(cand, min(c0 + (uint32_t)c, n_cand - 1));

  const RowView<DT> view = {static_cast<const T*>(x) + row * row_len};

  float a[4][NC];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < NC; ++c) a[j][c] = 0.0f;

  // Two CANDIDATES per VALU instruction (same x, packed fp32 pairs) with the exact branch-free quantizer of
  // tq_device.h (QF): per candidate-element  med3 + 1/2 (pk_mul + 2 pk_fma) + rndne + 1/2 (pk_mul, pk_add, pk_mul,
  // pk_add) = 5.5 issue slots, against ~12 plus a data-dependent branch for the guarded reciprocal.  The per-candidate
  // sum over the 16 steps of a chunk keeps its sequential order (the pair lanes are two different candidates).
  constexpr int NP = NC / 2;
  f32x2 sc2[NP], ns2[NP], rc2[NP];
  float ylo[NC], yhi[NC];
  bool fast = true;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const QF f = make_qf(QP{cp[c].scale, cp[c].zp, cp[c].lo, cp[c].hi});
    fast = fast && f.ok;
    ylo[c] = f.ylo;
    yhi[c] = f.yhi;
    if (c & 1) { sc2[c / 2].y = f.scale.x; ns2[c / 2].y = f.nscale.x; rc2[c / 2].y = f.rcp.x; }
    else { sc2[c / 2].x = f.scale.x; ns2[c / 2].x = f.nscale.x; rc2[c / 2].x = f.rcp.x; }
  }

  for (uint64_t t = t0; t < t1;) {
    const uint32_t n = (uint32_t)min((uint64_t)16, t1 - t);
    float xs[16];
    // padding with zeros is exact: 0 quantizes to 0 for every candidate, and s + 0 = s
    // (round 6: written as `j < n ? load : 0` alone, each of the 16 loads sits in its own exec-mask branch -- ~80 of the
    // ~260 instructions of a two-candidate chunk.  A full chunk in both half-waves is the usual case and loads
    // unconditionally: [8,128,768] x 100 candidates 40.0 -> 37.0 us, [256,512,768] x 100 1.97 -> 1.89 ms on one box,
    // profiles/r06/mse_uncond_ab.txt; a uniform-length variant for the short last chunk of 768-element rows changed
    // nothing, and requesting the next chunk before this one's arithmetic was slower: mse_prefetch_ab.txt.)
    if (__builtin_amdgcn_ballot_w64(n != 16) == 0) {      // a full chunk in both half-waves
#pragma unroll
      for (int j = 0; j < 16; ++j) xs[j] = view((t + j) * 32 + col);
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) xs[j] = (uint32_t)j < n ? view((t + j) * 32 + col) : 0.0f;
    }
    if (fast) {
      f32x2 s2[NP];
#pragma unroll
      for (int c = 0; c < NP; ++c) s2[c] = f32x2{a[0][2 * c], a[0][2 * c + 1]};
      // JB steps x NP candidate pairs = 4 independent dependency chains side by side, stage by stage in source order
      // (a dependent packed op needs a wait state; one chain per element made the scheduler emit an s_nop after every
      // packed instruction).  Narrow candidate tiles (NC = 4 / 2: few rows x few candidates, where 8 candidates per
      // wave leave most SIMDs idle) get their chains from consecutive steps instead; the accumulation into s2 keeps
      // its sequential step order either way.
      constexpr int JB = NP >= 4 ? 1 : 4 / NP;
      // (groups of 4 steps: a short last chunk -- rows of 768 elements are 16 + 8 steps -- skips its zero padding)
#pragma unroll
      for (int j0 = 0; j0 < 16; j0 += JB) {
        if ((j0 & 3) == 0 && j0 > 0 && (uint32_t)j0 >= n) break;
        f32x2 x2[JB], xc[JB][NP], q0[JB][NP], e[JB][NP], h[JB][NP], dl[JB][NP];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
          const float xv = xs[j0 + jj];
          x2[jj] = f32x2{xv, xv};
#pragma unroll
          for (int c = 0; c < NP; ++c) {
            xc[jj][c].x = __builtin_amdgcn_fmed3f(xv, ylo[2 * c], yhi[2 * c]);
            xc[jj][c].y = __builtin_amdgcn_fmed3f(xv, ylo[2 * c + 1], yhi[2 * c + 1]);
          }
        }
#pragma unroll
        for (int jj = 0; jj < JB; ++jj)
#pragma unroll
          for (int c = 0; c < NP; ++c) q0[jj][c] = xc[jj][c] * rc2[c];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj)
#pragma unroll
          for (int c = 0; c < NP; ++c) e[jj][c] = __builtin_elementwise_fma(q0[jj][c], ns2[c], xc[jj][c]);
#pragma unroll
        for (int jj = 0; jj < JB; ++jj)
#pragma unroll
          for (int c = 0; c < NP; ++c) q0[jj][c] = __builtin_elementwise_fma(e[jj][c], rc2[c], q0[jj][c]);
#pragma unroll
        for (int jj = 0; jj < JB; ++jj)
#pragma unroll
          for (int c = 0; c < NP; ++c) h[jj][c] = f32x2{rintf(q0[jj][c].x), rintf(q0[jj][c].y)};
#pragma unroll
        for (int jj = 0; jj < JB; ++jj)
#pragma unroll
          for (int c = 0; c < NP; ++c) dl[jj][c] = x2[jj] - sc2[c] * h[jj][c];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj)
#pragma unroll
          for (int c = 0; c < NP; ++c) dl[jj][c] = dl[jj][c] * dl[jj][c];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj)                       // sequential in step order, like the reference's sum
#pragma unroll
          for (int c = 0; c < NP; ++c) s2[c] = s2[c] + dl[jj][c];
      }
#pragma unroll
      for (int c = 0; c < NP; ++c) { a[0][2 * c] = s2[c].x; a[0][2 * c + 1] = s2[c].y; }
    } else {          // scales outside [2^-100, 2^100] / grids of 2^22+ steps: division path (rare: keep it small)
#pragma unroll 1
      for (int j = 0; j < 16; ++j) {
#pragma unroll
        for (int c = 0; c < NC; ++c) a[0][c] += sq_err(xs[j], cp[c]);
      }
    }
    t += n;
    if ((t & (L - 1)) == 0) TQ_ORD_DUMP(a, t, p, L, k_top, NC);
  }

  if (k_top < 4) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (c0 + c < n_cand) state[(ru * n_cand + c0 + c) * 32 + col] = make_float4(a[0][c], a[1][c], a[2][c], 0.0f);
    return;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const CandP pc = cp[c];
    auto val = [&](uint64_t e) { return sq_err(view(e), pc); };
    const float tot = row_len < 8 ? ordered_sum_small(row_len, val)
                                  : ordered_finish(a[0][c], a[1][c], a[2][c], a[3][c], row_len, steps, col, val);
    if (col == 0 && c0 + c < n_cand) row_loss[row * n_cand + c0 + c] = tot;
  }
}

// Stage 2 (rows split into units): replay the cascade levels >= k_top over the units of a row in order,
// then finish the row.  One half-wave per (row, candidate).
template <int DT>
__global__ __launch_bounds__(kBlock) void mse_ord_fold_k(const void* __restrict__ x, uint64_t row_len,
                                                         uint64_t steps, uint32_t p,
                                                         uint32_t k_top, uint64_t span, uint32_t units_per_row,
                                                         const float4* __restrict__ cand, uint32_t n_cand,
                                                         uint64_t n_rc, const float4* __restrict__ state,
                                                         float* __restrict__ row_loss) {
  typedef typename Store<DT>::elem_t T;
  const uint32_t col = threadIdx.x & 31u;
  const uint64_t rc = (uint64_t)blockIdx.x * (kBlock / 32) + (threadIdx.x >> 5);
  if (rc >= n_rc) return;
  const uint64_t row = rc / n_cand;
  const uint32_t c = (uint32_t)(rc - row * n_cand);
  const uint32_t L = 1u << p;
  const CandP pc = load_cand(cand, c);
  const RowView<DT> view = {static_cast<const T*>(x) + row * row_len};
  float acc2 = 0.0f, acc3 = 0.0f, A0 = 0.0f, A1 = 0.0f, A2 = 0.0f;
  for (uint32_t u = 0; u < units_per_row; ++u) {
    const uint64_t t0 = (uint64_t)u * span;
    const uint64_t t1 = min(steps, t0 + span);
    const float4 st = state[((row * units_per_row + u) * n_cand + c) * 32 + col];
    if (t1 - t0 == span) {
      if (k_top == 2) {
        acc2 += st.y;
        if (((t1 >> (2 * p)) & (L - 1)) == 0) { acc3 += acc2; acc2 = 0.0f; }
      } else {
        acc3 += st.z;
      }
    } else {
      A0 = st.x; A1 = st.y; A2 = st.z;
    }
  }
  const float l2 = k_top == 2 ? acc2 : A2;
  auto val = [&](uint64_t e) { return sq_err(view(e), pc); };
  const float tot = ordered_finish(A0, A1, l2, acc3, row_len, steps, col, val);
  if (col == 0) row_loss[rc] = tot;
}

// Stage 3: loss[c] += (double) ordered_sum_r row_loss[r, c]  (the reference's second torch.sum), or, for
// per-row losses (per_channel_loss=True), loss[r, c] += (double) row_loss[r, c].  One half-wave per candidate.
__global__ __launch_bounds__(kBlock) void mse_ord_rows_k(const float* __restrict__ row_loss, uint64_t rows,
                                                         uint32_t n_cand, uint32_t p, int reduce_rows,
                                                         double* __restrict__ loss, float* __restrict__ loss_f32) {
  const uint32_t col = threadIdx.x & 31u;
  const uint32_t c = blockIdx.x * (kBlock / 32) + (threadIdx.x >> 5);
  if (c >= n_cand) return;
  if (!reduce_rows) {
    for (uint64_t r = col; r < rows; r += 32) {
      const float v = row_loss[r * n_cand + c];
      if (loss) loss[r * n_cand + c] += (double)v;
      if (loss_f32) loss_f32[r * n_cand + c] = v;
    }
    return;
  }
  const PlainView view = {row_loss + c, n_cand};
  auto val = [&](uint64_t e) { return view(e); };
  float tot;
  if (rows < 8) {
    tot = ordered_sum_small(rows, val);
  } else {
    const uint64_t steps = (rows / 8) / 4;
    const uint32_t L = 1u << p;
    float a[4][1] = {{0.f}, {0.f}, {0.f}, {0.f}};
    for (uint64_t t = 0; t < steps;) {
      a[0][0] += val(t * 32 + col);
      ++t;
      if ((t & (L - 1)) == 0) TQ_ORD_DUMP(a, t, p, L, 4u, 1);
    }
    tot = ordered_finish(a[0][0], a[1][0], a[2][0], a[3][0], rows, steps, col, val);
  }
  if (col == 0) {
    if (loss) loss[c] += (double)tot;
    if (loss_f32) loss_f32[c] = tot;
  }
}

// ---------------------------------------------------------------------------------------------- host
static uint32_t ceil_log2_u64(uint64_t x) {     // c10::utils::CeilLog2
  if (x <= 2) return 1;
  uint32_t b = 0;
  for (uint64_t v = x - 1; v; v >>= 1) ++b;
  return b;
}
static uint32_t level_power(uint64_t steps) { return std::max<uint32_t>(4, ceil_log2_u64(steps) / 4); }

struct OrdPlan {
  uint64_t steps, span;
  uint32_t p, k_top, units_per_row, nc, n_ctiles;
  size_t state_bytes, row_loss_bytes;
};

static OrdPlan plan_ord(uint64_t rows, uint64_t row_len, uint64_t n_cand) {
  OrdPlan pl;
  pl.steps = (row_len / 8) / 4;
  pl.p = level_power(pl.steps);
  pl.nc = n_cand < 4 ? 2 : kOrdNC;
  pl.n_ctiles = (uint32_t)ceil_div(n_cand, pl.nc);
  const int forced_nc = tuning("TQ_ORD_NC", 0);            // tests / tuning: 2, 4 or 8
  const uint64_t L = 1ull << pl.p;
  // Largest unit that still gives >= 4 waves per SIMD (1024 SIMDs): splitting rows costs state traffic + the fold
  // kernel ([8,128,768] x 12800 candidates: 1.88 ms unsplit vs 2.11 ms split into 12 units per row).
  // TQ_ORD_KTOP forces a level (tests).
  const int forced = tuning("TQ_ORD_KTOP", 0);
  pl.k_top = 4;
  pl.span = std::max<uint64_t>(pl.steps, 1);
  pl.units_per_row = 1;
  for (uint32_t k = 4; k >= 2; --k) {
    uint64_t span = pl.span, upr = 1;
    if (k < 4) {
      span = k == 3 ? L * L * L : L * L;
      if (pl.steps <= span) continue;                     // a single unit either way: nothing to gain
      upr = ceil_div(pl.steps, span);
      if ((size_t)rows * upr * n_cand * 32 * sizeof(float4) > ((size_t)1 << 30)) break;
    }
    pl.k_top = k;
    pl.span = span;
    pl.units_per_row = (uint32_t)upr;
    if (forced ? (int)k <= forced : ceil_div(rows * upr, 2) * pl.n_ctiles >= 4096) break;
  }
  // Few rows x few candidates (the config shape [8, 98304] x 100: 624 waves of 8 candidates = 0.6 per SIMD, each a
  // serial chain of 256 steps x 8 candidates): narrower candidate tiles until there are ~2 waves per SIMD.  x is
  // re-read once per tile from L2 (3 MB tensor); the results do not depend on the tile width.
  const uint64_t items = ceil_div(rows * pl.units_per_row, 2);
  if (forced_nc == 2 || forced_nc == 4 || forced_nc == 8) pl.nc = (uint32_t)forced_nc;
  else
    while (pl.nc > 2 && items * ceil_div(n_cand, pl.nc) < 2048) pl.nc /= 2;
  pl.n_ctiles = (uint32_t)ceil_div(n_cand, pl.nc);
  pl.state_bytes = pl.k_top < 4 ? (size_t)rows * pl.units_per_row * n_cand * 32 * sizeof(float4) : 0;
  pl.row_loss_bytes = ((size_t)rows * n_cand * sizeof(float) + 15) & ~(size_t)15;
  return pl;
}

template <int DT>
static int launch_ord(const void* x, uint64_t rows, uint64_t row_len, const float* cand, uint64_t n_cand, int reduce_rows, double* loss, float* loss_f32, char* ws,
                      size_t ws_bytes, hipStream_t st) {
  const OrdPlan pl = plan_ord(rows, row_len, n_cand);
  const size_t need = pl.row_loss_bytes + pl.state_bytes;
  if (ws == nullptr || ws_bytes < need || !aligned16(ws))
    return set_error(TQ_EWORKSPACE, "tq_mse_candidates_ordered: workspace %zu < %zu (or not 16-byte aligned)", ws_bytes, need);
  float* row_loss = reinterpret_cast<float*>(ws);
  float4* state = reinterpret_cast<float4*>(ws + pl.row_loss_bytes);
  const float4* c4 = reinterpret_cast<const float4*>(cand);
  const uint64_t n_ru = rows * pl.units_per_row;
  const uint64_t waves = ceil_div(n_ru, 2) * pl.n_ctiles;
  const dim3 grid((unsigned)ceil_div(waves, kBlock / kWave));
  if (pl.nc == 2)
    hipLaunchKernelGGL((mse_ord_unit_k<DT, 2>), grid, dim3(kBlock), 0, st, x, row_len, pl.steps, pl.p,
                       pl.k_top, pl.span, pl.units_per_row, n_ru, c4, (uint32_t)n_cand, pl.n_ctiles, state, row_loss);
  else if (pl.nc == 4)
    hipLaunchKernelGGL((mse_ord_unit_k<DT, 4>), grid, dim3(kBlock), 0, st, x, row_len, pl.steps, pl.p,
                       pl.k_top, pl.span, pl.units_per_row, n_ru, c4, (uint32_t)n_cand, pl.n_ctiles, state, row_loss);
  else
    hipLaunchKernelGGL((mse_ord_unit_k<DT, kOrdNC>), grid, dim3(kBlock), 0, st, x, row_len, pl.steps,
                       pl.p, pl.k_top, pl.span, pl.units_per_row, n_ru, c4, (uint32_t)n_cand, pl.n_ctiles, state,
                       row_loss);
  if (int e = check_launch("mse_ord_unit_k")) return e;
  if (pl.k_top < 4) {
    const uint64_t n_rc = rows * n_cand;
    hipLaunchKernelGGL((mse_ord_fold_k<DT>), dim3((unsigned)ceil_div(n_rc, kBlock / 32)), dim3(kBlock), 0, st, x,
                       row_len, pl.steps, pl.p, pl.k_top, pl.span, pl.units_per_row, c4,
                       (uint32_t)n_cand, n_rc, state, row_loss);
    if (int e = check_launch("mse_ord_fold_k")) return e;
  }
  const uint32_t p_rows = level_power((rows / 8) / 4);
  hipLaunchKernelGGL(mse_ord_rows_k, dim3((unsigned)ceil_div(n_cand, kBlock / 32)), dim3(kBlock), 0, st, row_loss, rows,
                     (uint32_t)n_cand, p_rows, reduce_rows, loss, loss_f32);
  return check_launch("mse_ord_rows_k");
}

static int ord_dispatch(const void* x, uint64_t rows, uint64_t row_len, int dtype, const float* cand, uint64_t n_cand, int reduce_rows, double* loss, float* loss_f32,
                        void* workspace, size_t workspace_bytes, tq_stream_t stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  switch (dtype) {
    case TQ_F32: return launch_ord<TQ_F32>(x, rows, row_len, cand, n_cand, reduce_rows, loss, loss_f32, ws, workspace_bytes, st);
    case TQ_BF16: return launch_ord<TQ_BF16>(x, rows, row_len, cand, n_cand, reduce_rows, loss, loss_f32, ws, workspace_bytes, st);
    default: return launch_ord<TQ_F16>(x, rows, row_len, cand, n_cand, reduce_rows, loss, loss_f32, ws, workspace_bytes, st);
  }
}

}  // namespace tq

using namespace tq;

extern "C" size_t tq_mse_ordered_workspace_bytes(uint64_t rows, uint64_t row_len, uint64_t n_cand) {
  if (rows == 0 || n_cand == 0) return 0;
  const OrdPlan pl = plan_ord(rows, row_len, n_cand);
  return pl.row_loss_bytes + pl.state_bytes;
}

extern "C" int tq_mse_candidates_ordered(const void* x, uint64_t rows, uint64_t row_len, int dtype, const float* cand,
                                         uint64_t n_cand, int reduce_rows, double* loss, float* loss_f32,
                                         void* workspace, size_t workspace_bytes, tq_stream_t stream) {
  TQ_REQUIRE(x && cand && (loss || loss_f32), "tq_mse_candidates_ordered: NULL pointer");
  TQ_REQUIRE(rows >= 1 && row_len >= 1, "tq_mse_candidates_ordered: empty input");
  TQ_REQUIRE(n_cand >= 1 && n_cand < (1ull << 31), "tq_mse_candidates_ordered: bad candidate count");
  TQ_REQUIRE(dtype == TQ_F32 || dtype == TQ_BF16 || dtype == TQ_F16, "tq_mse_candidates_ordered: bad dtype %d", dtype);
  TQ_REQUIRE((reinterpret_cast<uintptr_t>(cand) & 15u) == 0, "tq_mse_candidates_ordered: candidate table must be 16-byte aligned");
  return ord_dispatch(x, rows, row_len, dtype, cand, n_cand, reduce_rows, loss, loss_f32, workspace,
                      workspace_bytes, stream);
}
