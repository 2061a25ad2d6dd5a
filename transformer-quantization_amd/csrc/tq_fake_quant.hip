// K1/K2/K3: fused quantize -> clip -> dequantize for gfx950 (MI355X).
//
// One pass over HBM: 16-byte vector loads (8 x bf16 / 4 x fp32 per lane), everything else in
// registers, 16-byte vector stores.  Replaces the 6 un-fused ATen kernels (~12 sweeps) behind
// AsymmetricUniformQuantizer.forward (reference quantization/quantizers.py:172-211).
//
// Kernel family
//   fq_tensor : one (scale, zp) for the whole tensor; parameters live in SGPRs.
//   fq_axis   : parameters indexed by the LAST axis (per-embedding / PEG); the per-column table
//               (scale, zp) is staged once per block in LDS; lanes read their 8 columns with
//               ds_read_b128.
//   fq_rows   : parameters constant along contiguous rows of `inner` elements (per-channel
//               weights, any non-last axis); one row per blockIdx.y, SGPR parameters.
//   fq_scalar : element-wise fallback for unaligned pointers / odd shapes.
#include <algorithm>

#include "tq_device.h"
#include "tq_host.h"

namespace tq {

template <int N, typename T> struct alignas(N * sizeof(T)) PackN { T e[N]; };

// (Round 6, measured and not kept -- profiles/r06/idx_pack_ab.txt: one v_cvt_pk_u8_f32 per element instead of convert + mask +
// shift/or for byte grids: bit-identical and 2 of ~10 instructions per element fewer, no change in any launch time --
// the index-only bf16 -> u8 launch (63-68 % of HBM) is not instruction-bound.)
template <int V>
__device__ __forceinline__ void store_idx(void* idx, int idx_dtype, uint64_t off, const float (&xi)[V]) {
  switch (idx_dtype) {  // wave-uniform
    case TQ_IDX_F32: { PackN<V, float> o;
#pragma unroll
      for (int j = 0; j < V; ++j) o.e[j] = xi[j];
      *reinterpret_cast<PackN<V, float>*>(static_cast<float*>(idx) + off) = o; break; }
    case TQ_IDX_I8: { PackN<V, int8_t> o;
#pragma unroll
      for (int j = 0; j < V; ++j) o.e[j] = (int8_t)(int)xi[j];
      *reinterpret_cast<PackN<V, int8_t>*>(static_cast<int8_t*>(idx) + off) = o; break; }
    case TQ_IDX_I8_M128: { PackN<V, int8_t> o;       // index - 128: the signed operand of the i8 GEMM
#pragma unroll
      for (int j = 0; j < V; ++j) o.e[j] = (int8_t)((int)xi[j] - 128);
      *reinterpret_cast<PackN<V, int8_t>*>(static_cast<int8_t*>(idx) + off) = o; break; }
    case TQ_IDX_U8: { PackN<V, uint8_t> o;
#pragma unroll
      for (int j = 0; j < V; ++j) o.e[j] = (uint8_t)(int)xi[j];
      *reinterpret_cast<PackN<V, uint8_t>*>(static_cast<uint8_t*>(idx) + off) = o; break; }
    case TQ_IDX_I16: { PackN<V, int16_t> o;
#pragma unroll
      for (int j = 0; j < V; ++j) o.e[j] = (int16_t)(int)xi[j];
      *reinterpret_cast<PackN<V, int16_t>*>(static_cast<int16_t*>(idx) + off) = o; break; }
    default: { PackN<V, int32_t> o;
#pragma unroll
      for (int j = 0; j < V; ++j) o.e[j] = (int32_t)xi[j];
      *reinterpret_cast<PackN<V, int32_t>*>(static_cast<int32_t*>(idx) + off) = o; break; }
  }
}

__device__ __forceinline__ void store_idx1(void* idx, int idx_dtype, uint64_t off, float xi) {
  switch (idx_dtype) {
    case TQ_IDX_F32: static_cast<float*>(idx)[off] = xi; break;
    case TQ_IDX_I8: static_cast<int8_t*>(idx)[off] = (int8_t)(int)xi; break;
    case TQ_IDX_I8_M128: static_cast<int8_t*>(idx)[off] = (int8_t)((int)xi - 128); break;
    case TQ_IDX_U8: static_cast<uint8_t*>(idx)[off] = (uint8_t)(int)xi; break;
    case TQ_IDX_I16: static_cast<int16_t*>(idx)[off] = (int16_t)(int)xi; break;
    default: static_cast<int32_t*>(idx)[off] = (int32_t)xi; break;
  }
}

// one 16-byte vector: widen, quantize, (store indices), dequantize, narrow
// FAST: the exact branch-free quantizer of tq_device.h (QF; ~7.5 issue slots per element incl. NaN propagation,
// data-independent) instead of the IEEE division (~19).  At 2 T elem/s (bf16 -> u8 index output, 3 B/elem) or
// 1.5 T elem/s (bf16 -> bf16) the division alone is 55-75 % of the VALU issue rate of the chip.
template <int DT, bool HAS_IDX, bool FAST = false>
__device__ __forceinline__ u32x4 fq_vec(const u32x4& in, const QP& p, void* idx, int idx_dtype, uint64_t elem_off,
                                        const QF* qf = nullptr) {
  constexpr int V = Store<DT>::kVec;
  float f[V];
  Store<DT>::unpack(in, f);
  if (FAST) {
    constexpr int H = V / 2;
    f32x2 x2[H], h[H];
#pragma unroll
    for (int j = 0; j < H; ++j) x2[j] = f32x2{f[2 * j], f[2 * j + 1]};
    qf_round2_n<H>(x2, *qf, h);
    const f32x2 zp2 = {qf->zp, qf->zp};
#pragma unroll
    for (int j = 0; j < H; ++j) {
      f32x2 xi = h[j] + zp2;
      xi.x = (f[2 * j] != f[2 * j]) ? f[2 * j] : xi.x;                   // torch.clamp keeps NaN; v_med3 does not
      xi.y = (f[2 * j + 1] != f[2 * j + 1]) ? f[2 * j + 1] : xi.y;
      x2[j] = xi;
    }
    if (HAS_IDX) {
#pragma unroll
      for (int j = 0; j < H; ++j) { f[2 * j] = x2[j].x; f[2 * j + 1] = x2[j].y; }
      store_idx<V>(idx, idx_dtype, elem_off, f);
    }
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const f32x2 yv = qf->scale * (x2[j] - zp2);
      f[2 * j] = yv.x;
      f[2 * j + 1] = yv.y;
    }
    return Store<DT>::pack(f);
  }
#pragma unroll
  for (int j = 0; j < V; ++j) f[j] = q_index(f[j], p);     // true division (scales outside [2^-100, 2^100], > 21 bits)
  if (HAS_IDX) store_idx<V>(idx, idx_dtype, elem_off, f);
#pragma unroll
  for (int j = 0; j < V; ++j) f[j] = q_dequant(f[j], p);
  return Store<DT>::pack(f);
}

// ------------------------------------------------------------------------------ per tensor
// Tile = kBlock * U consecutive 16-byte vectors (U = 4: 16 KiB in, 16 KiB out).  One tile per
// workgroup ("one shot": grid == number of tiles) measured 6.1-6.3 TB/s on MI355X against
// 4.4 TB/s for a 2048-block grid-stride loop over the same 805 MB tensor: with a persistent grid
// the resident blocks touch 2048 x U pages that lie 8 MB apart, the tile order keeps the
// resident set inside one contiguous ~32 MB window (DRAM page / TLB locality).  Blocks only loop
// when the tensor has more than kMaxTiles tiles.
constexpr unsigned kMaxTiles = 1u << 20;

// TPB consecutive tiles per block: 1 everywhere except 16-bit index-only output (2 B in, 1 B out per element), the one
// layout that gains from two (one box, profiles/r06/fq_tpb_ab.txt: 65.9 -> 69.4 % of HBM; the headline bf16 -> bf16
// launch loses: 76.9 -> 75.1 -> 71.0 % with 1 / 2 / 4).
template <int DT, bool HAS_IDX, bool NT, int U, bool FAST, int TPB = 1>
__device__ __forceinline__ void fq_tensor_tiles(const u32x4* __restrict__ x, u32x4* __restrict__ y,
                                                void* __restrict__ idx, int idx_dtype, uint64_t n_vec, const QP& p,
                                                const QF& qf) {
  constexpr int V = Store<DT>::kVec;
  constexpr uint64_t TILE = (uint64_t)kBlock * U;
  for (uint64_t tb = (uint64_t)blockIdx.x * (TILE * TPB); tb < n_vec; tb += (uint64_t)gridDim.x * (TILE * TPB))
#pragma unroll 1
  for (uint64_t t0 = tb; t0 < tb + TILE * TPB && t0 < n_vec; t0 += TILE) {
    const uint64_t i = t0 + threadIdx.x;
    if (t0 + TILE <= n_vec) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = NT ? ld_stream(x + i + u * kBlock) : x[i + u * kBlock];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const u32x4 o = fq_vec<DT, HAS_IDX, FAST>(v[u], p, idx, idx_dtype, (i + u * kBlock) * V, &qf);
        if (y) { if (NT) st_stream(y + i + u * kBlock, o); else y[i + u * kBlock] = o; }
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t k = i + u * kBlock;
        if (k < n_vec) {
          const u32x4 o = fq_vec<DT, HAS_IDX, FAST>(x[k], p, idx, idx_dtype, k * V, &qf);
          if (y) y[k] = o;
        }
      }
    }
  }
}

template <int DT, bool HAS_IDX, bool NT, int U, int TPB = 1>
__global__ __launch_bounds__(kBlock) void fq_tensor(const u32x4* __restrict__ x, u32x4* __restrict__ y,
                                                    void* __restrict__ idx, int idx_dtype, uint64_t n,
                                                    tq_quantizer q) {
  constexpr int V = Store<DT>::kVec;
  const QP p = make_qp(q, 0);
  const QF qf = make_qf(p);
  const uint64_t n_vec = n / V;
  if (qf.ok) fq_tensor_tiles<DT, HAS_IDX, NT, U, true, TPB>(x, y, idx, idx_dtype, n_vec, p, qf);
  else fq_tensor_tiles<DT, HAS_IDX, NT, U, false, TPB>(x, y, idx, idx_dtype, n_vec, p, qf);
  // ragged tail (< V elements)
  const uint64_t tail0 = n_vec * V;
  if (blockIdx.x == 0 && tail0 + threadIdx.x < n) {
    typedef typename Store<DT>::elem_t E;
    const uint64_t k = tail0 + threadIdx.x;
    const float xi = q_index(Store<DT>::load1(reinterpret_cast<const E*>(x) + k), p);
    if (HAS_IDX) store_idx1(idx, idx_dtype, k, xi);
    if (y) Store<DT>::store1(reinterpret_cast<E*>(y) + k, q_dequant(xi, p));
  }
}

// ------------------------------------------------------------------------------ per tensor, parameters from statistics
// The second half of a sharded per-tensor calibrating step (tq_calibrate_apply) without its own update launch: the
// estimator rule (range_estimators.py:83-216) and range -> parameters (quantizers.py:258-259, 276-277, 335-339) are a
// dozen scalar operations, so every block redoes them from the all-reduced [-min, max] and the previous state, in the
// exact operation order of calib_update_k (tq_stats.hip), and quantizes its tile; block 0 stores the new state.
template <int DT, bool NT, int U>
__global__ __launch_bounds__(kBlock) void fq_tensor_calib(const u32x4* __restrict__ x, u32x4* __restrict__ y, uint64_t n,
                                                          CalibApplyArgs c) {
  constexpr int V = Store<DT>::kVec;
  float a, b;
  if (c.partials != nullptr) {
    // single-GPU step: the statistics launch left one (min, max) pair per block; every block folds the <= 512 pairs
    // itself (4 KB from L2) -- no last-block ticket, no device-scope fence in the statistics kernel
    __shared__ float s_mm[2][kBlock / kWave];
    float mn = __builtin_huge_valf(), mx = -__builtin_huge_valf();
    for (uint32_t t = threadIdx.x; t < c.n_partials; t += kBlock) {
      mn = min_nanprop(mn, c.partials[2 * t]);
      mx = max_nanprop(mx, c.partials[2 * t + 1]);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if ((threadIdx.x & (kWave - 1)) == 0) { s_mm[0][threadIdx.x / kWave] = mn; s_mm[1][threadIdx.x / kWave] = mx; }
    __syncthreads();
    a = s_mm[0][0];
    b = s_mm[1][0];
#pragma unroll
    for (int k = 1; k < kBlock / kWave; ++k) { a = min_nanprop(a, s_mm[0][k]); b = max_nanprop(b, s_mm[1][k]); }
  } else {
    a = -c.stats[0];
    b = c.stats[1];
  }
  if (!(c.mode == TQ_EST_CURRENT || c.prev_min == nullptr)) {
    const float pa = c.prev_min[0], pb = c.prev_max[0];
    if (c.mode == TQ_EST_ALL) { a = min_nanprop(pa, a); b = max_nanprop(pb, b); }
    else { a = c.om * a + c.mom * pa; b = c.om * b + c.mom * pb; }
  }
  const float lo = min_nanprop(a, 0.0f), hi = max_nanprop(b, c.eps);
  QP p;
  float d_store, zf_store = 0.0f;
  bool sgn = false;
  if (c.symmetric) {
    sgn = lo < 0.0f;
    const float d = max_nanprop(fabsf(lo), hi) / grid_top(c.n_bits - (sgn ? 1 : 0));
    d_store = c.log_domain ? logf(d) : d;
    p.zp = 0.0f;
    p.lo = sgn ? -(float)ldexp(1.0, c.n_bits - 1) : 0.0f;
    p.hi = grid_top(c.n_bits - (sgn ? 1 : 0));
  } else {
    const float d = (hi - lo) / grid_top(c.n_bits);
    zf_store = (-lo) / d;
    d_store = c.log_domain ? logf(d) : d;
    p.lo = 0.0f;
    p.hi = grid_top(c.n_bits);
    p.zp = clamp_nanprop(rintf(zf_store), p.lo, p.hi);
  }
  p.scale = c.log_domain ? expf(d_store) : (d_store < c.eps ? c.eps : d_store);      // make_qp on the stored values
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    c.cur_min[0] = a;
    c.cur_max[0] = b;
    c.delta[0] = d_store;
    if (c.symmetric) c.signed_flag[0] = sgn ? 1 : 0;
    else c.zero_float[0] = zf_store;
  }
  const QF qf = make_qf(p);
  const uint64_t n_vec = n / V;
  if (qf.ok) fq_tensor_tiles<DT, false, NT, U, true>(x, y, nullptr, TQ_IDX_NONE, n_vec, p, qf);
  else fq_tensor_tiles<DT, false, NT, U, false>(x, y, nullptr, TQ_IDX_NONE, n_vec, p, qf);
  const uint64_t tail0 = n_vec * V;
  if (blockIdx.x == 0 && tail0 + threadIdx.x < n) {
    typedef typename Store<DT>::elem_t E;
    const uint64_t k = tail0 + threadIdx.x;
    const float xi = q_index(Store<DT>::load1(reinterpret_cast<const E*>(x) + k), p);
    Store<DT>::store1(reinterpret_cast<E*>(y) + k, q_dequant(xi, p));
  }
}

// ------------------------------------------------------------------------------ last axis
// x viewed as [rows, d]; d % V == 0.  LDS: scale[d], zp[d].  Same tiling as fq_tensor, but a block
// owns TPB consecutive tiles so that the table fill (6 KB for d = 768) is paid once per 64 KiB of
// input, and the first tile's HBM loads are issued BEFORE the fill + barrier so their latency
// overlaps it.  The vector column of lane/slot (tid, u) is (tile_start + tid + u * kBlock) mod (d / V),
// kept incrementally.
template <int DT, bool HAS_IDX, bool NT, int U>
__global__ __launch_bounds__(kBlock) void fq_axis(const u32x4* __restrict__ x, u32x4* __restrict__ y,
                                                  void* __restrict__ idx, int idx_dtype, uint64_t n,
                                                  tq_quantizer q, uint32_t tpb) {
  constexpr int V = Store<DT>::kVec;
  constexpr uint32_t TILE = kBlock * U;
  extern __shared__ __attribute__((aligned(16))) float s_par[];
  const uint32_t d = (uint32_t)q.n_params;
  float* s_scale = s_par;
  float* s_zp = s_par + d;
  float* s_rcp = s_par + 2 * d;                     // guarded reciprocals (tq_device.h)
  const uint64_t n_vec = n / V;
  const uint32_t vpr = d / V;                       // vectors per row (<= 2048)

  auto load_tile = [&](uint64_t tile, u32x4 (&v)[U]) {
    const uint64_t i0 = tile * TILE + threadIdx.x;
    if ((tile + 1) * TILE <= n_vec) {           // full tile: no per-lane predicates
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = NT ? ld_stream(x + i0 + (uint64_t)u * kBlock) : x[i0 + (uint64_t)u * kBlock];
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t k = i0 + (uint64_t)u * kBlock;
        v[u] = u32x4{0, 0, 0, 0};
        if (k < n_vec) v[u] = x[k];
      }
    }
  };

  const uint64_t n_tiles = (n_vec + TILE - 1) / TILE;
  uint64_t tile = (uint64_t)blockIdx.x * tpb;
  u32x4 v[U];
  if (tile < n_tiles) load_tile(tile, v);

  for (uint32_t c = threadIdx.x; c < d; c += kBlock) {
    const QP p = make_qp(q, c);
    // "plane" layout: floats j = 4m .. 4m+3 of every vector column sit in plane m, so a ds_read_b128 has a
    // lane stride of 16 B (conflict-free: 16 lanes x 16 B = all 64 banks); with the natural layout a bf16
    // lane's 8 parameters are 32 B apart from its neighbour's and every read is 2-way conflicted
    const uint32_t slot = ((c % V) / 4 * vpr + c / V) * 4 + (c & 3);
    s_scale[slot] = p.scale;
    s_zp[slot] = p.zp;
    s_rcp[slot] = guarded_rcp(p.scale);
  }
  const QP p0 = make_qp(q, 0);   // int_min / int_max do not depend on the column
  const float lo = p0.lo, hi = p0.hi;
  __syncthreads();

  const uint32_t tid_mod = threadIdx.x % vpr;
  const uint32_t blk_mod = kBlock % vpr;
  const uint32_t tile_mod = TILE % vpr;

  for (; tile < n_tiles; tile += (uint64_t)gridDim.x * tpb) {
    for (uint32_t t = 0; t < tpb && tile + t < n_tiles; ++t) {
      const uint64_t cur = tile + t;
      if (t > 0 || cur != (uint64_t)blockIdx.x * tpb) load_tile(cur, v);
      const uint64_t i0 = cur * TILE + threadIdx.x;
      const bool full = (cur + 1) * TILE <= n_vec;
      uint32_t cv = ((uint32_t)(cur % vpr) * tile_mod) % vpr + tid_mod;
      if (cv >= vpr) cv -= vpr;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t k = i0 + (uint64_t)u * kBlock;
        float g[V], f[V], sc[V], zp[V];
        Store<DT>::unpack(v[u], g);
#pragma unroll
        for (int j = 0; j < V; j += 4) {
          const f32x4 r4 = *reinterpret_cast<const f32x4*>(s_rcp + ((j / 4) * vpr + cv) * 4);
          const f32x4 s4 = *reinterpret_cast<const f32x4*>(s_scale + ((j / 4) * vpr + cv) * 4);
          const f32x4 z4 = *reinterpret_cast<const f32x4*>(s_zp + ((j / 4) * vpr + cv) * 4);
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            sc[j + m] = s4[m];
            zp[j + m] = z4[m];
            f[j + m] = clamp_nanprop(rne_quot1(g[j + m], s4[m], r4[m]) + z4[m], lo, hi);
          }
        }
        if (full || k < n_vec) {
          if (HAS_IDX) store_idx<V>(idx, idx_dtype, k * V, f);
          if (y) {
#pragma unroll
            for (int j = 0; j < V; ++j) f[j] = sc[j] * (f[j] - zp[j]);
            const u32x4 o = Store<DT>::pack(f);
            if (NT) st_stream(y + k, o); else y[k] = o;
          }
        }
        cv += blk_mod;
        if (cv >= vpr) cv -= vpr;
      }
    }
  }
}

// Register-resident variant: the block size is chosen as a multiple of the vectors per row, so a lane
// owns ONE vector column for its whole life and keeps that column's parameters in registers: no LDS,
// no barrier, no column arithmetic in the loop.  Tile = blockDim.x * U consecutive vectors; a block takes `tpb`
// consecutive tiles (round 6: it took one, so the V parameter derivations per lane -- an IEEE division each -- cost as
// much as the quantization of its U vectors; and the element math is now the exact branch-free form of tq_device.h with
// per-column constants: med3 against the column's dequantised grid ends, Markstein quotient on register pairs, instead
// of the guarded reciprocal with its data-dependent branch per element).
template <int DT, bool HAS_IDX, bool NT, int U, int MAXB>
__global__ __launch_bounds__(MAXB) void fq_axis_reg(const u32x4* __restrict__ x, u32x4* __restrict__ y,
                                                    void* __restrict__ idx, int idx_dtype, uint64_t n, tq_quantizer q,
                                                    uint32_t tpb) {
  constexpr int V = Store<DT>::kVec;
  constexpr int H = V / 2;
  const uint32_t bs = blockDim.x;
  const uint64_t tile_vecs = (uint64_t)bs * U;
  const uint64_t n_vec = n / V;
  const uint32_t vpr = (uint32_t)q.n_params / V;
  const uint64_t n_tiles = (n_vec + tile_vecs - 1) / tile_vecs;
  uint64_t tile = (uint64_t)blockIdx.x * tpb;
  if (tile >= n_tiles) return;
  const uint64_t tile_end = tile + tpb < n_tiles ? tile + tpb : n_tiles;

  u32x4 v[U];
  auto load_tile = [&](uint64_t t) {
    const uint64_t i0 = t * tile_vecs + threadIdx.x;
    if ((t + 1) * tile_vecs <= n_vec) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = NT ? ld_stream(x + i0 + (uint64_t)u * bs) : x[i0 + (uint64_t)u * bs];
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t k = i0 + (uint64_t)u * bs;
        v[u] = u32x4{0, 0, 0, 0};
        if (k < n_vec) v[u] = x[k];
      }
    }
  };
  load_tile(tile);                                  // data loads in flight while the parameters arrive

  // parameters of this lane's V columns: straight 16-byte loads first, arithmetic afterwards (make_qp per
  // column would serialise 2 V dependent scalar loads behind its wave-uniform branches)
  const uint32_t c0 = (threadIdx.x % vpr) * V;
  float sc[V], zp[V];
  {
    f32x4 dv[V / 4], zv[V / 4];
#pragma unroll
    for (int j = 0; j < V / 4; ++j) {
      dv[j] = *reinterpret_cast<const f32x4*>(q.delta + c0 + 4 * j);
      zv[j] = q.symmetric ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(q.zero_float + c0 + 4 * j);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) { sc[j] = dv[j / 4][j % 4]; zp[j] = zv[j / 4][j % 4]; }
  }
  const QP p0 = make_qp(q, 0);                        // int_min / int_max do not depend on the column
  const float lo = p0.lo, hi = p0.hi;
  float rc[V], ylo[V], yhi[V];
  bool ok = true;                                   // every column admits the exact branch-free form (QF::ok)
#pragma unroll
  for (int j = 0; j < V; ++j) {
    sc[j] = q.log_domain ? expf(sc[j]) : (sc[j] < q.eps ? q.eps : sc[j]);              // quantizers.py:142-147
    zp[j] = q.symmetric ? 0.0f : clamp_nanprop(rintf(zp[j]), lo, hi);                  // :149-153, :330-332
    rc[j] = guarded_rcp(sc[j]);
    const float klo = lo - zp[j], khi = hi - zp[j];
    ylo[j] = sc[j] * klo;
    yhi[j] = sc[j] * khi;
    ok = ok && (rc[j] == rc[j]) && fabsf(klo) < 4194304.0f && fabsf(khi) < 4194304.0f;
  }
  for (;;) {
    const uint64_t i0 = tile * tile_vecs + threadIdx.x;
    const bool full = (tile + 1) * tile_vecs <= n_vec;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t k = i0 + (uint64_t)u * bs;
      float g[V], f[V];
      Store<DT>::unpack(v[u], g);
      if (ok) {
        f32x2 xc[H], q0[H], e[H];
#pragma unroll
        for (int j = 0; j < H; ++j) {
          xc[j].x = __builtin_amdgcn_fmed3f(g[2 * j], ylo[2 * j], yhi[2 * j]);
          xc[j].y = __builtin_amdgcn_fmed3f(g[2 * j + 1], ylo[2 * j + 1], yhi[2 * j + 1]);
        }
#pragma unroll
        for (int j = 0; j < H; ++j) q0[j] = xc[j] * f32x2{rc[2 * j], rc[2 * j + 1]};
#pragma unroll
        for (int j = 0; j < H; ++j) e[j] = __builtin_elementwise_fma(q0[j], f32x2{-sc[2 * j], -sc[2 * j + 1]}, xc[j]);
#pragma unroll
        for (int j = 0; j < H; ++j) q0[j] = __builtin_elementwise_fma(e[j], f32x2{rc[2 * j], rc[2 * j + 1]}, q0[j]);
#pragma unroll
        for (int j = 0; j < H; ++j) {
          const f32x2 xi = f32x2{rintf(q0[j].x), rintf(q0[j].y)} + f32x2{zp[2 * j], zp[2 * j + 1]};
          f[2 * j] = (g[2 * j] != g[2 * j]) ? g[2 * j] : xi.x;                        // torch.clamp keeps NaN; v_med3 does not
          f[2 * j + 1] = (g[2 * j + 1] != g[2 * j + 1]) ? g[2 * j + 1] : xi.y;
        }
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) f[j] = clamp_nanprop(rne_quot1(g[j], sc[j], rc[j]) + zp[j], lo, hi);
      }
      if (full || k < n_vec) {
        if (HAS_IDX) store_idx<V>(idx, idx_dtype, k * V, f);
        if (y) {
#pragma unroll
          for (int j = 0; j < H; ++j) {
            const f32x2 yv = f32x2{sc[2 * j], sc[2 * j + 1]} * (f32x2{f[2 * j], f[2 * j + 1]} - f32x2{zp[2 * j], zp[2 * j + 1]});
            f[2 * j] = yv.x;
            f[2 * j + 1] = yv.y;
          }
          const u32x4 o = Store<DT>::pack(f);
          if (NT) st_stream(y + k, o); else y[k] = o;
        }
      }
    }
    if (++tile >= tile_end) break;
    load_tile(tile);
  }
}

// ------------------------------------------------------------------------------ affine + quant
// MobileBERT's NoNorm followed by its output quantizer (reference models/quantized_mobilebert.py:
// 58-72): y = Q(x * w[col] + b[col]) with a per-tensor quantizer.  The reference runs mul, add and
// the 6 quantizer kernels as separate sweeps (~16 tensor passes); here it is 1 read + 1 write.
// x viewed as [rows, d], d % V == 0; w, b (fp32 [d], already fake-quantized) staged in LDS.
template <int DT, bool NT, int U>
__global__ __launch_bounds__(kBlock) void fq_affine(const u32x4* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ b, u32x4* __restrict__ y,
                                                    int8_t* __restrict__ y_idx, uint64_t n, uint32_t d, tq_quantizer q) {
  constexpr int V = Store<DT>::kVec;
  constexpr uint32_t TILE = kBlock * U;
  extern __shared__ __attribute__((aligned(16))) float s_par[];
  float* s_w = s_par;
  float* s_b = s_par + d;
  const uint64_t n_vec = n / V;
  const uint32_t vpr = d / V;
  const uint64_t tile = blockIdx.x;
  const uint64_t i0 = tile * TILE + threadIdx.x;
  const bool full = (tile + 1) * TILE <= n_vec;
  u32x4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint64_t k = i0 + (uint64_t)u * kBlock;
    v[u] = u32x4{0, 0, 0, 0};
    if (full || k < n_vec) v[u] = NT ? ld_stream(x + k) : x[k];
  }
  for (uint32_t c = threadIdx.x; c < d; c += kBlock) {      // plane layout, as in fq_axis (conflict-free b128 reads)
    const uint32_t slot = ((c % V) / 4 * vpr + c / V) * 4 + (c & 3);
    s_w[slot] = w[c];
    s_b[slot] = b[c];
  }
  const QP p = make_qp(q, 0);
  __syncthreads();
  uint32_t cv = ((uint32_t)(tile % vpr) * (TILE % vpr)) % vpr + threadIdx.x % vpr;
  if (cv >= vpr) cv -= vpr;
  const uint32_t blk_mod = kBlock % vpr;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint64_t k = i0 + (uint64_t)u * kBlock;
    float f[V];
    struct alignas(V) { int8_t e[V]; } oi;
    Store<DT>::unpack(v[u], f);
#pragma unroll
    for (int j = 0; j < V; j += 4) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(s_w + ((j / 4) * vpr + cv) * 4);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(s_b + ((j / 4) * vpr + cv) * 4);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float r = f[j + m] * w4[m] + b4[m];          // mul, then add (no fma: -ffp-contract=off)
        const float xi = q_index(r, p);
        oi.e[j + m] = (int8_t)((int)xi - 128);             // only stored for asymmetric <= 8-bit quantizers (host check)
        f[j + m] = q_dequant(xi, p);
      }
    }
    if (full || k < n_vec) {
      const u32x4 o = Store<DT>::pack(f);
      if (NT) st_stream(y + k, o); else y[k] = o;
      if (y_idx != nullptr) {                             // int8(index - 128) for a following integer Linear
        if constexpr (V == 4) *reinterpret_cast<uint32_t*>(y_idx + k * V) = __builtin_bit_cast(uint32_t, oi);
        else                  *reinterpret_cast<u32x2*>(y_idx + k * V) = __builtin_bit_cast(u32x2, oi);
      }
    }
    cv += blk_mod;
    if (cv >= vpr) cv -= vpr;
  }
}

// ------------------------------------------------------------------------------ rows
// x viewed as [n_rows, inner]; parameter index = row % n_params; inner % V == 0.
template <int DT, bool HAS_IDX>
__global__ __launch_bounds__(kBlock) void fq_rows(const u32x4* __restrict__ x, u32x4* __restrict__ y,
                                                  void* __restrict__ idx, int idx_dtype, uint64_t n_rows,
                                                  tq_quantizer q) {
  constexpr int V = Store<DT>::kVec;
  const uint64_t vec_per_row = q.inner / V;
  for (uint64_t row = blockIdx.y; row < n_rows; row += gridDim.y) {
    const QP p = make_qp(q, row % q.n_params);
    const uint64_t base = row * vec_per_row;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < vec_per_row;
         i += (uint64_t)gridDim.x * kBlock) {
      const u32x4 o = fq_vec<DT, HAS_IDX>(x[base + i], p, idx, idx_dtype, (base + i) * V);
      if (y) y[base + i] = o;
    }
  }
}

// ------------------------------------------------------------------------------ short rows
// x viewed as [outer, n_params, inner], rows of at most kWaveRowMaxVec vectors (per-token ranges of a [B, T, d]
// activation, reference main.py:359-376; per-channel weights).  fq_rows gives every row a 256-thread block (96 lanes
// busy for d = 768 bf16) and derives the parameters once per row in every lane.  Here wave (p, s) owns the rows
// o = s, s + S, ... of parameter p: the parameters are derived ONCE per wave (wave-uniform), the body is the per-tensor
// one, and at any time the resident waves sweep one contiguous window of S x n_params rows (see mm_rows_wave).
template <int DT, bool HAS_IDX, bool NT, bool FAST>
__device__ __forceinline__ void fq_rows_wave_body(const u32x4* __restrict__ x, u32x4* __restrict__ y, void* __restrict__ idx,
                                                  int idx_dtype, uint64_t outer, uint32_t vpr, uint32_t n_params, uint32_t S,
                                                  uint32_t p_at, uint32_t s, const QP& p, const QF& qf) {
  constexpr int V = Store<DT>::kVec;
  constexpr int U = 4;
  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint64_t row_stride = (uint64_t)n_params * vpr;
  const uint64_t base = (uint64_t)p_at * vpr;
  uint64_t o = s;
  // (tried in round 5 and slower, profiles/r05/fq_rows_seq_ab.txt: the U rows of a batch as one flat index space over the
  // lanes -- all lanes busy, 6-8 loads in flight: 59 % against 66 %; sequential tiles with an LDS parameter table: 60 %)
  for (; o + (uint64_t)(U - 1) * S < outer; o += (uint64_t)U * S) {
    for (uint32_t i = lane; i < vpr; i += kWave) {
      u32x4 v[U];
      uint64_t at[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        at[u] = base + (o + (uint64_t)u * S) * row_stride + i;
        v[u] = NT ? ld_stream(x + at[u]) : x[at[u]];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const u32x4 r = fq_vec<DT, HAS_IDX, FAST>(v[u], p, idx, idx_dtype, at[u] * V, &qf);
        if (y) { if (NT) st_stream(y + at[u], r); else y[at[u]] = r; }
      }
    }
  }
  for (; o < outer; o += S) {
    for (uint32_t i = lane; i < vpr; i += kWave) {
      const uint64_t at = base + o * row_stride + i;
      const u32x4 r = fq_vec<DT, HAS_IDX, FAST>(x[at], p, idx, idx_dtype, at * V, &qf);
      if (y) y[at] = r;
    }
  }
}

template <int DT, bool HAS_IDX, bool NT>
__global__ __launch_bounds__(kBlock) void fq_rows_wave(const u32x4* __restrict__ x, u32x4* __restrict__ y,
                                                       void* __restrict__ idx, int idx_dtype, uint64_t outer, uint32_t vpr,
                                                       uint32_t S, tq_quantizer q) {
  const uint32_t n_params = (uint32_t)q.n_params;
  const uint64_t gw = (uint64_t)blockIdx.x * (kBlock / kWave) + __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  if (gw >= (uint64_t)n_params * S) return;
  const uint32_t p_at = (uint32_t)(gw % n_params), s = (uint32_t)(gw / n_params);
  const QP p = make_qp(q, p_at);
  const QF qf = make_qf(p);
  if (qf.ok) fq_rows_wave_body<DT, HAS_IDX, NT, true>(x, y, idx, idx_dtype, outer, vpr, n_params, S, p_at, s, p, qf);
  else fq_rows_wave_body<DT, HAS_IDX, NT, false>(x, y, idx, idx_dtype, outer, vpr, n_params, S, p_at, s, p, qf);
}

// ------------------------------------------------------------------------------ per row, flat tile order (round 6)
// The same quantizers as fq_rows_wave -- parameters constant along rows of `inner` contiguous elements: per-token
// activations (reference main.py:359-376), per-channel weights -- in the tile order of fq_tensor: a workgroup owns
// kBlock * U CONSECUTIVE 16-byte vectors, whatever rows they belong to, and every lane derives the parameters of the row
// its vector lies in.  fq_rows_wave gives a wave one token position and walks the batch: four streams of 1.5-3 KB rows
// that lie S x T rows apart (66 % of HBM at [1024,512,768] for bf16 AND for fp32, whose rows fill every lane); here the
// resident blocks sweep one contiguous window like the per-tensor kernel, at the price of ~45 more instructions per
// vector (two exact divisions by launch constants, three cached 4-byte parameter reads, make_qp + make_qf per lane),
// which an HBM-bound kernel has room for (~200 issue slots per vector at 6 TB/s).
//
// row = floor(k / vpr), parameter = row mod n_params, both through a double-precision reciprocal: for a < 2^32 the
// product a * RN(1 / d) is within a * 2^-52 < 2^-20 of a / d, so its truncation is floor(a / d) or -- only when d
// divides a -- one less; one comparison of the remainder repairs that.
__device__ __forceinline__ uint32_t div_exact_u32(uint32_t a, uint32_t d, double rd, uint32_t& rem) {
  uint32_t qt = (uint32_t)((double)a * rd);
  uint32_t r = a - qt * d;
  if (r >= d) { r -= d; qt += 1; }
  rem = r;
  return qt;
}

template <int DT, bool HAS_IDX, bool NT, int U>
__global__ __launch_bounds__(kBlock) void fq_rows_flat(const u32x4* __restrict__ x, u32x4* __restrict__ y,
                                                       void* __restrict__ idx, int idx_dtype, uint64_t n_vec, uint32_t vpr,
                                                       double rcp_vpr, double rcp_np, tq_quantizer q) {
  constexpr int V = Store<DT>::kVec;
  constexpr uint64_t TILE = (uint64_t)kBlock * U;
  const uint32_t n_params = (uint32_t)q.n_params;
  for (uint64_t t0 = (uint64_t)blockIdx.x * TILE; t0 < n_vec; t0 += (uint64_t)gridDim.x * TILE) {
    const uint64_t i = t0 + threadIdx.x;
    const bool full = t0 + TILE <= n_vec;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t k = i + (uint64_t)u * kBlock;
      if (full || k < n_vec) v[u] = NT ? ld_stream(x + k) : x[k];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t k = i + (uint64_t)u * kBlock;
      if (full || k < n_vec) {
        uint32_t rem;
        const uint32_t row = div_exact_u32((uint32_t)k, vpr, rcp_vpr, rem);
        div_exact_u32(row, n_params, rcp_np, rem);
        const QP p = make_qp<true>(q, rem);
        const QF qf = make_qf(p);
        u32x4 o;
        if (qf.ok) o = fq_vec<DT, HAS_IDX, true>(v[u], p, idx, idx_dtype, k * V, &qf);
        else o = fq_vec<DT, HAS_IDX, false>(v[u], p, idx, idx_dtype, k * V, &qf);
        if (y) { if (NT) st_stream(y + k, o); else y[k] = o; }
      }
    }
  }
}

// The same with the parameters of the rows a tile touches derived ONCE per tile into LDS (rows of >= 4 vectors: a tile of
// kBlock * U vectors touches at most kBlock * U / 4 + 2 rows): per vector a local row number (one fma + truncation: exact for
// offsets below 2^22, see the comment in the body), two broadcast LDS reads and the quantizer itself.  The per-lane form
// above costs ~45 instructions per vector on top of the quantizer: 78 % of HBM for fp32 but 67 % for bf16 (8 elements per
// vector) at [1024,512,768], and slower than the wave-per-row kernel on tensors of a few MB (profiles/r06/rows_flat_ab.txt).
struct RowQ { float scale, rcp, ylo, yhi, zp, lo, hi, ok; };

template <int DT, bool HAS_IDX, bool NT, int U>
__global__ __launch_bounds__(kBlock) void fq_rows_tab(const u32x4* __restrict__ x, u32x4* __restrict__ y,
                                                      void* __restrict__ idx, int idx_dtype, uint64_t n_vec, uint32_t vpr,
                                                      double rcp_vpr, double rcp_np, float rcp_vpr_f, tq_quantizer q) {
  constexpr int V = Store<DT>::kVec;
  constexpr uint32_t TILE = kBlock * U;
  __shared__ __attribute__((aligned(16))) RowQ s_q[TILE / 4 + 2];
  const uint32_t n_params = (uint32_t)q.n_params;
  const float half_rcp = 0.5f * rcp_vpr_f;
  for (uint64_t t0 = (uint64_t)blockIdx.x * TILE; t0 < n_vec; t0 += (uint64_t)gridDim.x * TILE) {
    const uint64_t i = t0 + threadIdx.x;
    const bool full = t0 + TILE <= n_vec;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t k = i + (uint64_t)u * kBlock;
      if (full || k < n_vec) v[u] = NT ? ld_stream(x + k) : x[k];
    }
    uint32_t rem0;
    const uint32_t row0 = div_exact_u32((uint32_t)t0, vpr, rcp_vpr, rem0);      // first row of the tile, offset inside it
    for (uint32_t r = threadIdx.x; (uint64_t)r * vpr < (uint64_t)rem0 + TILE; r += kBlock) {
      uint32_t pi;
      div_exact_u32(row0 + r, n_params, rcp_np, pi);
      const QP p = make_qp<true>(q, pi);
      const QF qf = make_qf(p);
      s_q[r] = RowQ{p.scale, qf.rcp.x, qf.ylo, qf.yhi, p.zp, p.lo, p.hi, qf.ok ? 1.0f : 0.0f};
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t k = i + (uint64_t)u * kBlock;
      if (full || k < n_vec) {
        // local row = floor(j / vpr), j = offset from the start of the tile's first row (< vpr + TILE < 2^22):
        // (j + 0.5) / vpr lies >= 0.5 / vpr away from every integer, the computed value within (j + 0.5) / vpr * 2^-23 of it
        const uint32_t j = rem0 + (uint32_t)u * kBlock + threadIdx.x;
        const uint32_t rl = (uint32_t)__builtin_fmaf((float)j, rcp_vpr_f, half_rcp);
        const RowQ e = s_q[rl];
        const QP p = {e.scale, e.zp, e.lo, e.hi};
        QF qf;
        qf.scale = f32x2{e.scale, e.scale};
        qf.nscale = f32x2{-e.scale, -e.scale};
        qf.rcp = f32x2{e.rcp, e.rcp};
        qf.ylo = e.ylo; qf.yhi = e.yhi; qf.zp = e.zp; qf.ok = e.ok != 0.0f;
        u32x4 o;
        if (qf.ok) o = fq_vec<DT, HAS_IDX, true>(v[u], p, idx, idx_dtype, k * V, &qf);
        else o = fq_vec<DT, HAS_IDX, false>(v[u], p, idx, idx_dtype, k * V, &qf);
        if (y) { if (NT) st_stream(y + k, o); else y[k] = o; }
      }
    }
    if (t0 + (uint64_t)gridDim.x * TILE < n_vec) __syncthreads();    // the table is rewritten by the next tile
  }
}

// ------------------------------------------------------------------------------ fallback
template <int DT, bool HAS_IDX>
__global__ __launch_bounds__(kBlock) void fq_scalar(const void* __restrict__ x, void* __restrict__ y,
                                                    void* __restrict__ idx, int idx_dtype, uint64_t n,
                                                    tq_quantizer q) {
  typedef typename Store<DT>::elem_t E;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const uint64_t pi = q.n_params == 1 ? 0 : (i / q.inner) % q.n_params;
    const QP p = make_qp(q, pi);
    const float xi = q_index(Store<DT>::load1(static_cast<const E*>(x) + i), p);
    if (HAS_IDX) store_idx1(idx, idx_dtype, i, xi);
    if (y) Store<DT>::store1(static_cast<E*>(y) + i, q_dequant(xi, p));
  }
}

template <int DT, bool HAS_IDX>
static int launch_fq(const void* x, void* y, void* idx, int idx_dtype, uint64_t n, const tq_quantizer& q,
                     hipStream_t st) {
  constexpr int V = Store<DT>::kVec;
  const size_t idx_es = idx_dtype == TQ_IDX_F32 || idx_dtype == TQ_IDX_I32 ? 4 : (idx_dtype == TQ_IDX_I16 ? 2 : 1);   // I8 / U8 / I8_M128: 1
  const bool vec_ok = aligned16(x) && (y == nullptr || aligned16(y)) &&
                      (!HAS_IDX || (reinterpret_cast<uintptr_t>(idx) % (V * idx_es)) == 0);
  static const int nt_min_mb = tuning("TQ_NT_MIN_MB", 64);   // streaming hint above this footprint
  const bool nt = (n * elem_size(DT)) >= ((uint64_t)nt_min_mb << 20);
  const auto xv = static_cast<const u32x4*>(x);
  auto yv = static_cast<u32x4*>(y);

  // tile size: U = 4 vectors per lane once there are enough tiles to fill the chip, else 1
  const uint64_t n_vec_all = n / V;
  const bool big = n_vec_all >= (uint64_t)kBlock * 4 * 2048;
  if (vec_ok && q.n_params == 1) {
#define TQ_LAUNCH_TENSOR(NTV, UV)                                                                          \
    hipLaunchKernelGGL((fq_tensor<DT, HAS_IDX, NTV, UV>),                                                   \
                       dim3((unsigned)std::min<uint64_t>(std::max<uint64_t>(ceil_div(n_vec_all, kBlock * UV), 1), kMaxTiles)), \
                       dim3(kBlock), 0, st, xv, yv, idx, idx_dtype, n, q)
    if (HAS_IDX && DT != TQ_F32 && y == nullptr && big && nt) {     // 16-bit index-only at size: two tiles per block
      hipLaunchKernelGGL((fq_tensor<DT, HAS_IDX, true, 4, HAS_IDX ? 2 : 1>),
                         dim3((unsigned)std::min<uint64_t>(ceil_div(n_vec_all, kBlock * 4 * 2), kMaxTiles)), dim3(kBlock), 0, st, xv, yv,
                         idx, idx_dtype, n, q);
      return check_launch("fq_tensor");
    }
    if (big) { if (nt) TQ_LAUNCH_TENSOR(true, 4); else TQ_LAUNCH_TENSOR(false, 4); }
    else     { if (nt) TQ_LAUNCH_TENSOR(true, 1); else TQ_LAUNCH_TENSOR(false, 1); }
#undef TQ_LAUNCH_TENSOR
    return check_launch("fq_tensor");
  }
  // per-embedding parameters, row of vpr = d / V vectors: block size = a multiple of vpr (<= 1024)
  // (measured: the register variant wins for fp32 rows of <= 256 vectors, the LDS-table variant for bf16)
  static const int axis_reg = tuning("TQ_AXIS_REG", 2);   // 0: LDS table first, 1: registers first, 2: by dtype
  const bool lds_ok = vec_ok && q.n_params > 1 && q.inner == 1 && q.n_params % V == 0 && q.n_params <= 5440;
  const bool reg_ok = vec_ok && q.n_params > 1 && q.inner == 1 && q.n_params % V == 0 && q.n_params / V <= 1024 &&
                      aligned16(q.delta) && (q.symmetric || aligned16(q.zero_float));
  // Round 6 (profiles/r06/axis_wide_ab.txt, axis_reg_tpb.txt; one box each): with several tiles per block and the packed
  // exact element math the register kernel is ahead for fp32 at every width (d = 768: 76 % against 69 % of HBM at
  // [1024,512,768]; d = 3072: 75 % against 65 %, and 9 us against 20 us at [8,128,3072]) and for 16-bit tensors with an
  // output on wide rows or large launches (d = 768: 77 % against 73 %; d = 3072: 68 % against 60 %); the LDS table stays
  // for 16-bit launches of a few MB on rows of <= 1024 columns and for 16-bit index-only output.
  const bool prefer_reg = axis_reg == 1 || (axis_reg == 2 && (DT == TQ_F32 || (y != nullptr && (q.n_params >= 2048 || big))));
  if (reg_ok && (prefer_reg || !lds_ok)) {
    const uint32_t vpr = (uint32_t)(q.n_params / V);
    const uint32_t bs = vpr <= kBlock ? vpr * (kBlock / vpr) : vpr;
    static const int reg_tpb = tuning("TQ_AXIS_REG_TPB", 0);
#define TQ_LAUNCH_AXIS_REG(NTV, UV, MB)                                                                    \
    {                                                                                                      \
      const uint64_t n_tiles = std::max<uint64_t>(ceil_div(n_vec_all, (uint64_t)bs * UV), 1);               \
      /* consecutive tiles per block: U x tpb vectors per lane pay for its V parameter derivations -- 16 for 16-bit  \
         storage (V = 8), 4 for fp32, twice that without an output; more than that costs locality (the resident blocks  \
         then sweep a window tpb times larger: fp32 [1024,512,768] 76 % with one tile, 66 % with eight) */  \
      const uint64_t rule = (uint64_t)(DT == TQ_F32 ? 4 : 16) / UV * (y != nullptr ? 1 : 2);                  \
      const uint32_t tpb = reg_tpb > 0 ? (uint32_t)reg_tpb : (uint32_t)std::max<uint64_t>(std::min<uint64_t>(rule, n_tiles / 2048), 1); \
      hipLaunchKernelGGL((fq_axis_reg<DT, HAS_IDX, NTV, UV, MB>), dim3((unsigned)ceil_div(n_tiles, tpb)),   \
                         dim3(bs), 0, st, xv, yv, idx, idx_dtype, n, q, tpb);                               \
    }
    if (bs <= kBlock) {
      if (big) { if (nt) TQ_LAUNCH_AXIS_REG(true, 4, kBlock) else TQ_LAUNCH_AXIS_REG(false, 4, kBlock) }
      else     { if (nt) TQ_LAUNCH_AXIS_REG(true, 1, kBlock) else TQ_LAUNCH_AXIS_REG(false, 1, kBlock) }
    } else {
      if (big) { if (nt) TQ_LAUNCH_AXIS_REG(true, 2, 1024) else TQ_LAUNCH_AXIS_REG(false, 2, 1024) }
      else     { if (nt) TQ_LAUNCH_AXIS_REG(true, 1, 1024) else TQ_LAUNCH_AXIS_REG(false, 1, 1024) }
    }
#undef TQ_LAUNCH_AXIS_REG
    return check_launch("fq_axis_reg");
  }
  if (lds_ok) {   // 3 x d floats of LDS <= 64 KiB
    const size_t lds = q.n_params * 3 * sizeof(float);
    static const int tpb_env = tuning("TQ_AXIS_TPB", 0);
#define TQ_LAUNCH_AXIS(NTV, UV)                                                                            \
    {                                                                                                      \
      const uint64_t n_tiles = std::max<uint64_t>(ceil_div(n_vec_all, kBlock * UV), 1);                    \
      const uint32_t want = q.n_params > 2048 ? 8 : (q.n_params > 1024 ? 4 : 2);  /* amortise the table fill */ \
      const uint32_t tpb = tpb_env > 0 ? (uint32_t)tpb_env : (n_tiles >= 2048ull * want ? want : (n_tiles >= 4096 ? 2 : 1));          \
      hipLaunchKernelGGL((fq_axis<DT, HAS_IDX, NTV, UV>),                                                   \
                         dim3((unsigned)std::min<uint64_t>(ceil_div(n_tiles, tpb), kMaxTiles)),            \
                         dim3(kBlock), lds, st, xv, yv, idx, idx_dtype, n, q, tpb);                         \
    }
    if (big) { if (nt) TQ_LAUNCH_AXIS(true, 4) else TQ_LAUNCH_AXIS(false, 4) }
    else     { if (nt) TQ_LAUNCH_AXIS(true, 1) else TQ_LAUNCH_AXIS(false, 1) }
#undef TQ_LAUNCH_AXIS
    return check_launch("fq_axis");
  }
  // rows of `inner` elements sharing a parameter: the flat-tile kernels once the launch moves >= 192 MB (measured on one
  // box, profiles/r06/rows_flat_ab.txt: [1024,512,768] 66 -> 73-76 % of HBM for bf16, 67-76 -> 77-78 % for fp32, ahead
  // from [128,512,768] bf16 / [64,512,768] fp32 upwards; below that -- tensors that stay in the 256 MB infinity cache from
  // one launch to the next -- the wave-per-row kernel is ahead, and for 16-bit index-only output at every size).
  // TQ_ROWS_FLAT=0: never, 1: by size, 2: the per-lane form everywhere, 3: the table form everywhere.
  const int rows_flat = tuning("TQ_ROWS_FLAT", 1);     // (read per launch: tests/test_per_token.py switches it)
  const uint64_t moved = n * elem_size(DT) * (y != nullptr ? 2 : 1);
  const bool flat_pays = rows_flat >= 2 || (moved >= (192ull << 20) && (y != nullptr || DT == TQ_F32));
  if (rows_flat && flat_pays && vec_ok && q.n_params > 1 && q.inner > 1 && q.inner % V == 0 && n_vec_all < (1ull << 32) && n % V == 0 &&
      q.n_params < (1ull << 32)) {
    const uint32_t vpr = (uint32_t)(q.inner / V);
    const double rcp_vpr = 1.0 / (double)vpr, rcp_np = 1.0 / (double)q.n_params;
#define TQ_LAUNCH_ROWS_FLAT(NTV, UV)                                                                       \
    hipLaunchKernelGGL((fq_rows_flat<DT, HAS_IDX, NTV, UV>),                                                \
                       dim3((unsigned)std::min<uint64_t>(std::max<uint64_t>(ceil_div(n_vec_all, kBlock * UV), 1), kMaxTiles)), \
                       dim3(kBlock), 0, st, xv, yv, idx, idx_dtype, n_vec_all, vpr, rcp_vpr, rcp_np, q)
#define TQ_LAUNCH_ROWS_TAB(NTV, UV)                                                                        \
    hipLaunchKernelGGL((fq_rows_tab<DT, HAS_IDX, NTV, UV>),                                                 \
                       dim3((unsigned)std::min<uint64_t>(std::max<uint64_t>(ceil_div(n_vec_all, kBlock * UV), 1), kMaxTiles)), \
                       dim3(kBlock), 0, st, xv, yv, idx, idx_dtype, n_vec_all, vpr, rcp_vpr, rcp_np, (float)rcp_vpr, q)
    if (vpr >= 4 && vpr <= (1u << 20) && rows_flat != 2) {       // (TQ_ROWS_FLAT=2: the per-lane form everywhere)
      if (big) { if (nt) TQ_LAUNCH_ROWS_TAB(true, 4); else TQ_LAUNCH_ROWS_TAB(false, 4); }
      else     { if (nt) TQ_LAUNCH_ROWS_TAB(true, 1); else TQ_LAUNCH_ROWS_TAB(false, 1); }
      return check_launch("fq_rows_tab");
    }
#undef TQ_LAUNCH_ROWS_TAB
    if (big) { if (nt) TQ_LAUNCH_ROWS_FLAT(true, 4); else TQ_LAUNCH_ROWS_FLAT(false, 4); }
    else     { if (nt) TQ_LAUNCH_ROWS_FLAT(true, 1); else TQ_LAUNCH_ROWS_FLAT(false, 1); }
#undef TQ_LAUNCH_ROWS_FLAT
    return check_launch("fq_rows_flat");
  }
  if (vec_ok && q.n_params > 1 && q.inner > 1 && q.inner % V == 0 && q.inner / V <= kWaveRowMaxVec &&
      q.n_params <= (1u << 24) && n % (q.n_params * q.inner) == 0) {
    const uint64_t outer = n / (q.n_params * q.inner);
    const uint32_t S = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(ceil_div(kWaveRowTarget, q.n_params), outer));
    const unsigned gx = (unsigned)ceil_div(q.n_params * S, (uint64_t)(kBlock / kWave));
    if (nt) hipLaunchKernelGGL((fq_rows_wave<DT, HAS_IDX, true>), dim3(gx), dim3(kBlock), 0, st, xv, yv, idx, idx_dtype, outer,
                               (uint32_t)(q.inner / V), S, q);
    else    hipLaunchKernelGGL((fq_rows_wave<DT, HAS_IDX, false>), dim3(gx), dim3(kBlock), 0, st, xv, yv, idx, idx_dtype, outer,
                               (uint32_t)(q.inner / V), S, q);
    return check_launch("fq_rows_wave");
  }
  if (vec_ok && q.n_params > 1 && q.inner > 1 && q.inner % V == 0) {
    const uint64_t n_rows = n / q.inner;
    const uint64_t vpr = q.inner / V;
    const unsigned gx = (unsigned)std::min<uint64_t>(std::max<uint64_t>(ceil_div(vpr, kBlock), 1), 64);
    const unsigned gy = (unsigned)std::min<uint64_t>(n_rows, 65535);
    hipLaunchKernelGGL((fq_rows<DT, HAS_IDX>), dim3(gx, gy), dim3(kBlock), 0, st, xv, yv, idx, idx_dtype, n_rows, q);
    return check_launch("fq_rows");
  }
  const unsigned grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>(ceil_div(n, kBlock), 1), kMaxGrid);
  hipLaunchKernelGGL((fq_scalar<DT, HAS_IDX>), dim3(grid), dim3(kBlock), 0, st, x, y, idx, idx_dtype, n, q);
  return check_launch("fq_scalar");
}

template <int DT>
static int launch_calib_fq(const void* x, void* y, uint64_t n, const CalibApplyArgs& c, hipStream_t st) {
  constexpr int V = Store<DT>::kVec;
  static const int nt_min_mb = tuning("TQ_NT_MIN_MB", 64);
  const bool nt = (n * elem_size(DT)) >= ((uint64_t)nt_min_mb << 20);
  const uint64_t n_vec_all = n / V;
  const bool big = n_vec_all >= (uint64_t)kBlock * 4 * 2048;
  const auto xv = static_cast<const u32x4*>(x);
  auto yv = static_cast<u32x4*>(y);
#define TQ_LAUNCH_CALIB(NTV, UV)                                                                           \
  hipLaunchKernelGGL((fq_tensor_calib<DT, NTV, UV>),                                                        \
                     dim3((unsigned)std::min<uint64_t>(std::max<uint64_t>(ceil_div(n_vec_all, kBlock * UV), 1), kMaxTiles)), \
                     dim3(kBlock), 0, st, xv, yv, n, c)
  if (big) { if (nt) TQ_LAUNCH_CALIB(true, 4); else TQ_LAUNCH_CALIB(false, 4); }
  else     { if (nt) TQ_LAUNCH_CALIB(true, 1); else TQ_LAUNCH_CALIB(false, 1); }
#undef TQ_LAUNCH_CALIB
  return check_launch("fq_tensor_calib");
}

int launch_fq_from_stats(const void* x, void* y, uint64_t n, int dtype, const CalibApplyArgs& c, hipStream_t st) {
  switch (dtype) {
    case TQ_F32: return launch_calib_fq<TQ_F32>(x, y, n, c, st);
    case TQ_BF16: return launch_calib_fq<TQ_BF16>(x, y, n, c, st);
    default: return launch_calib_fq<TQ_F16>(x, y, n, c, st);
  }
}

// ------------------------------------------------------------------------------ dynamic step, one pass
// Block p owns parameter p of an [outer, n_params, inner] tensor whose outer * inner / V vectors fit in the block's
// registers (MAXV per lane): load once, block min / max (torch's NaN propagation), estimator rule and range -> parameters in
// the operation order of calib_update_k (tq_stats.hip; reference range_estimators.py:83-216, quantizers.py:258-259, 276-277),
// make_qp on the STORED values, quantize from registers, store.  1 read + 1 write of x instead of 2 reads + 1 write, one
// launch instead of four: `--dynamic --per-token` (reference main.py:249, 359-376) runs this on every inference call.
template <int DT, int MAXV>
__global__ __launch_bounds__(kBlock) void calib_rows_onepass_k(const u32x4* __restrict__ x, u32x4* __restrict__ y,
                                                               uint32_t outer, uint32_t vpr, RowsOnePassArgs a) {
  constexpr int V = Store<DT>::kVec;
  const uint32_t p = blockIdx.x, n_params = (uint32_t)a.n_params;
  const uint32_t total = outer * vpr;
  u32x4 v[MAXV];
  uint64_t at[MAXV];
  MinMax acc;
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    const uint32_t j = threadIdx.x + u * kBlock;
    if (j < total) {
      const uint32_t o = j / vpr, i = j - o * vpr;
      at[u] = ((uint64_t)o * n_params + p) * vpr + i;
      v[u] = x[at[u]];
    }
  }
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    if (threadIdx.x + u * kBlock < total) {
      float f[V];
      Store<DT>::unpack(v[u], f);
#pragma unroll
      for (int k = 0; k < V; ++k) acc.add(f[k]);
    }
  }
  __shared__ float s_mm[2][kBlock / kWave];
  float mn = wave_min(acc.lo()), mx = wave_max(acc.hi());
  if ((threadIdx.x & (kWave - 1)) == 0) { s_mm[0][threadIdx.x / kWave] = mn; s_mm[1][threadIdx.x / kWave] = mx; }
  __syncthreads();
  float ea = s_mm[0][0], eb = s_mm[1][0];
#pragma unroll
  for (int k = 1; k < kBlock / kWave; ++k) { ea = min_nanprop(ea, s_mm[0][k]); eb = max_nanprop(eb, s_mm[1][k]); }
  if (!(a.mode == TQ_EST_CURRENT || a.prev_min == nullptr)) {
    const float pa = a.prev_min[p], pb = a.prev_max[p];
    if (a.mode == TQ_EST_ALL) { ea = min_nanprop(pa, ea); eb = max_nanprop(pb, eb); }
    else { ea = a.om * ea + a.mom * pa; eb = a.om * eb + a.mom * pb; }
  }
  const float lo = min_nanprop(ea, 0.0f), hi = max_nanprop(eb, a.eps);
  const float top = grid_top(a.n_bits);
  const float d = (hi - lo) / top;
  const float zf_store = (-lo) / d;
  const float d_store = a.log_domain ? logf(d) : d;
  QP qp;
  qp.lo = 0.0f;
  qp.hi = top;
  qp.zp = clamp_nanprop(rintf(zf_store), qp.lo, qp.hi);
  qp.scale = a.log_domain ? expf(d_store) : (d_store < a.eps ? a.eps : d_store);
  __syncthreads();                       // every thread has read prev_* (the state may be updated in place)
  if (threadIdx.x == 0) {
    a.cur_min[p] = ea;
    a.cur_max[p] = eb;
    a.delta[p] = d_store;
    a.zero_float[p] = zf_store;
  }
  const QF qf = make_qf(qp);
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    if (threadIdx.x + u * kBlock < total) {
      const u32x4 r = qf.ok ? fq_vec<DT, false, true>(v[u], qp, nullptr, TQ_IDX_NONE, 0, &qf)
                            : fq_vec<DT, false, false>(v[u], qp, nullptr, TQ_IDX_NONE, 0, &qf);
      y[at[u]] = r;
    }
  }
}

int launch_calib_rows_onepass(const void* x, void* y, int dtype, const RowsOnePassArgs& a, hipStream_t st) {
  static const int enabled = tuning("TQ_DYN_ONEPASS", 1);
  const uint64_t V = dtype == TQ_F32 ? 4 : 8;
  constexpr int MAXV = 8;
  if (!enabled || a.n_params < 2 || a.inner < 2 || a.inner % V || !aligned16(x) || !aligned16(y) || a.n_params > (1u << 20))
    return -1;
  const uint64_t vpr = a.inner / V, total = a.outer * vpr;
  if (total > (uint64_t)kBlock * MAXV || a.outer > 0xffffu) return -1;
  const auto xv = static_cast<const u32x4*>(x);
  auto yv = static_cast<u32x4*>(y);
#define TQ_GO(DTV, MV)                                                                                          \
  hipLaunchKernelGGL((calib_rows_onepass_k<DTV, MV>), dim3((unsigned)a.n_params), dim3(kBlock), 0, st, xv, yv,  \
                     (uint32_t)a.outer, (uint32_t)vpr, a)
#define TQ_PICK(DTV)                                                            \
  if (total <= (uint64_t)kBlock * 2) TQ_GO(DTV, 2);                             \
  else if (total <= (uint64_t)kBlock * 4) TQ_GO(DTV, 4);                        \
  else TQ_GO(DTV, 8)
  switch (dtype) {
    case TQ_F32: TQ_PICK(TQ_F32); break;
    case TQ_BF16: TQ_PICK(TQ_BF16); break;
    default: TQ_PICK(TQ_F16); break;
  }
#undef TQ_PICK
#undef TQ_GO
  return check_launch("calib_rows_onepass_k");
}

// ------------------------------------------------------------------------------ STE backward
// dx = ((g * scale) * mask) / scale      (autograd of mul / clamp / STE-round / div in order)
// d_delta, d_zero_float (per-tensor only): chain rule through scale = clamp(delta, eps),
//   zp = clamp(round_ste(zero_float), lo, hi), x_int = clamp(round_ste(x/s) + zp, lo, hi),
//   y = s * (x_int - zp).
struct BwdAcc { float d, z; };

__device__ __forceinline__ float ste_bwd_elem(float xv, float g, const QP& p, bool pgrad, BwdAcc& acc) {
  const float r = rintf(xv / p.scale) + p.zp;
  const bool in = (r >= p.lo) && (r <= p.hi);          // torch.clamp backward mask (inclusive)
  const float gs = g * p.scale;                         // grad wrt (x_int - zp)
  if (pgrad) {
    const float xi = clamp_nanprop(r, p.lo, p.hi);
    // d y / d scale = (x_int - zp) - mask * x / s ;  d y / d zp = s * (mask - 1)
    float dd = g * (xi - p.zp);
    if (in) dd -= (gs * xv) / (p.scale * p.scale);
    acc.d += dd;
    acc.z += in ? 0.0f : -gs;
  }
  return in ? gs / p.scale : 0.0f;
}

__device__ __forceinline__ void bwd_block_reduce(BwdAcc acc, float* __restrict__ partial) {
  __shared__ float s_red[2][kBlock / kWave];
  acc.d = wave_sum(acc.d);
  acc.z = wave_sum(acc.z);
  const int w = threadIdx.x / kWave;
  if ((threadIdx.x & (kWave - 1)) == 0) { s_red[0][w] = acc.d; s_red[1][w] = acc.z; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float d = 0.f, z = 0.f;
    for (int k = 0; k < kBlock / kWave; ++k) { d += s_red[0][k]; z += s_red[1][k]; }
    partial[2 * blockIdx.x] = d;
    partial[2 * blockIdx.x + 1] = z;
  }
}

// per-tensor, vectorised: 3 streams (x, g in; gx out) = 6 B/elem bf16; same one-shot tiling as K1
template <int DT, bool PGRAD, bool NT, int U>
__global__ __launch_bounds__(kBlock) void fq_bwd_tensor(const u32x4* __restrict__ x, const u32x4* __restrict__ gy,
                                                        u32x4* __restrict__ gx, float* __restrict__ partial,
                                                        uint64_t n, tq_quantizer q) {
  constexpr int V = Store<DT>::kVec;
  constexpr uint64_t TILE = (uint64_t)kBlock * U;
  const QP p = make_qp(q, 0);
  const uint64_t n_vec = n / V;
  BwdAcc acc = {0.f, 0.f};
  for (uint64_t t0 = (uint64_t)blockIdx.x * TILE; t0 < n_vec; t0 += (uint64_t)gridDim.x * TILE) {
    const uint64_t i = t0 + threadIdx.x;
    const bool full = t0 + TILE <= n_vec;
    u32x4 vx[U], vg[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t k = i + u * kBlock;
      vx[u] = u32x4{0, 0, 0, 0};
      vg[u] = u32x4{0, 0, 0, 0};
      if (full || k < n_vec) { vx[u] = NT ? ld_stream(x + k) : x[k]; vg[u] = NT ? ld_stream(gy + k) : gy[k]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t k = i + u * kBlock;
      float fx[V], fg[V];
      Store<DT>::unpack(vx[u], fx);
      Store<DT>::unpack(vg[u], fg);
      if (full || k < n_vec) {
#pragma unroll
        for (int j = 0; j < V; ++j) fg[j] = ste_bwd_elem(fx[j], fg[j], p, PGRAD, acc);
        const u32x4 o = Store<DT>::pack(fg);
        if (NT) st_stream(gx + k, o); else gx[k] = o;
      }
    }
  }
  const uint64_t tail0 = n_vec * V;
  if (blockIdx.x == 0 && tail0 + threadIdx.x < n) {
    typedef typename Store<DT>::elem_t E;
    const uint64_t k = tail0 + threadIdx.x;
    const float r = ste_bwd_elem(Store<DT>::load1(reinterpret_cast<const E*>(x) + k),
                                 Store<DT>::load1(reinterpret_cast<const E*>(gy) + k), p, PGRAD, acc);
    Store<DT>::store1(reinterpret_cast<E*>(gx) + k, r);
  }
  if (PGRAD) bwd_block_reduce(acc, partial);
}

// any layout / alignment (scalar loads); parameter gradients only for per-tensor quantizers
template <int DT, bool PGRAD>
__global__ __launch_bounds__(kBlock) void fq_bwd(const void* __restrict__ x, const void* __restrict__ gy,
                                                 void* __restrict__ gx, float* __restrict__ partial, uint64_t n,
                                                 tq_quantizer q) {
  typedef typename Store<DT>::elem_t E;
  BwdAcc acc = {0.f, 0.f};
  const bool per_tensor = q.n_params == 1;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const QP p = make_qp(q, per_tensor ? 0 : (i / q.inner) % q.n_params);
    const float r = ste_bwd_elem(Store<DT>::load1(static_cast<const E*>(x) + i),
                                 Store<DT>::load1(static_cast<const E*>(gy) + i), p, PGRAD && per_tensor, acc);
    Store<DT>::store1(static_cast<E*>(gx) + i, r);
  }
  if (PGRAD) bwd_block_reduce(acc, partial);
}

// sum the block partials and apply the chain rule through scale = clamp(delta, eps) | exp(delta)
// and zp = clamp(round_ste(zero_float), lo, hi)
__global__ void fq_bwd_final(const float* __restrict__ partial, uint32_t nb, tq_quantizer q, float* __restrict__ g_delta,
                             float* __restrict__ g_zf) {
  __shared__ double s_red[2][kBlock / kWave];
  double d = 0.0, z = 0.0;
  for (uint32_t i = threadIdx.x; i < nb; i += kBlock) { d += (double)partial[2 * i]; z += (double)partial[2 * i + 1]; }
  d = wave_sum(d);
  z = wave_sum(z);
  const int w = threadIdx.x / kWave;
  if ((threadIdx.x & (kWave - 1)) == 0) { s_red[0][w] = d; s_red[1][w] = z; }
  __syncthreads();
  if (threadIdx.x == 0) {
    d = z = 0.0;
    for (int k = 0; k < kBlock / kWave; ++k) { d += s_red[0][k]; z += s_red[1][k]; }
    const float delta = q.delta[0];
    const float pass = q.log_domain ? expf(delta) : (delta >= q.eps ? 1.0f : 0.0f);
    g_delta[0] = (float)d * pass;
    if (g_zf != nullptr && !q.symmetric) {
      const QP p = make_qp(q, 0);
      const float zf = rintf(q.zero_float[0]);
      g_zf[0] = (zf >= p.lo && zf <= p.hi) ? (float)z : 0.0f;
    }
  }
}

// ---- parameter gradients for per-channel / per-axis ranges (learn_ranges with vector parameters) ----------------
// x viewed as [outer, n_params, inner].  Block (p, s) reduces parameter p over slice s of the `outer` index
// (deterministic: block partials + a final kernel, no atomics).  The element gradient gx comes from fq_bwd.
constexpr unsigned kBwdSlices = 64;
__host__ __device__ inline unsigned bwd_param_slices(uint64_t outer, uint64_t inner) {
  const uint64_t per_param = outer * inner;
  uint64_t s = per_param / 4096;
  if (s < 1) s = 1;
  if (s > kBwdSlices) s = kBwdSlices;
  if (s > outer) s = outer;
  return (unsigned)s;
}

template <int DT>
__global__ __launch_bounds__(kBlock) void fq_bwd_params_k(const void* __restrict__ x, const void* __restrict__ gy,
                                                          uint64_t outer, tq_quantizer q, float* __restrict__ partial) {
  typedef typename Store<DT>::elem_t E;
  __shared__ float s_red[2][kBlock / kWave];
  const uint64_t prm = blockIdx.x;
  const QP p = make_qp(q, prm);
  const uint64_t chunk = (outer + gridDim.y - 1) / gridDim.y;
  const uint64_t o0 = (uint64_t)blockIdx.y * chunk, o1 = min(outer, o0 + chunk);
  BwdAcc acc = {0.f, 0.f};
  const uint64_t per = (o1 > o0 ? o1 - o0 : 0) * q.inner;
  for (uint64_t e = threadIdx.x; e < per; e += kBlock) {
    const uint64_t o = o0 + e / q.inner, i = e % q.inner;
    const uint64_t at = (o * q.n_params + prm) * q.inner + i;
    ste_bwd_elem(Store<DT>::load1(static_cast<const E*>(x) + at), Store<DT>::load1(static_cast<const E*>(gy) + at), p, true,
                 acc);
  }
  acc.d = wave_sum(acc.d);
  acc.z = wave_sum(acc.z);
  const int w = threadIdx.x / kWave;
  if ((threadIdx.x & (kWave - 1)) == 0) { s_red[0][w] = acc.d; s_red[1][w] = acc.z; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float d = 0.f, z = 0.f;
    for (int k = 0; k < kBlock / kWave; ++k) { d += s_red[0][k]; z += s_red[1][k]; }
    partial[(prm * gridDim.y + blockIdx.y) * 2] = d;
    partial[(prm * gridDim.y + blockIdx.y) * 2 + 1] = z;
  }
}

// per parameter: sum the slices (fixed order, fp64) and apply the chain rule through scale = clamp(delta, eps) | exp(delta)
// and zp = clamp(round_ste(zero_float), lo, hi)
__global__ void fq_bwd_params_final_k(const float* __restrict__ partial, uint32_t slices, tq_quantizer q,
                                      float* __restrict__ g_delta, float* __restrict__ g_zf) {
  const uint64_t prm = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (prm >= q.n_params) return;
  double d = 0.0, z = 0.0;
  for (uint32_t s = 0; s < slices; ++s) { d += (double)partial[(prm * slices + s) * 2]; z += (double)partial[(prm * slices + s) * 2 + 1]; }
  const float delta = q.delta[prm];
  const float pass = q.log_domain ? expf(delta) : (delta >= q.eps ? 1.0f : 0.0f);
  g_delta[prm] = (float)d * pass;
  if (g_zf != nullptr && !q.symmetric) {
    const QP p = make_qp(q, prm);
    const float zf = rintf(q.zero_float[prm]);
    g_zf[prm] = (zf >= p.lo && zf <= p.hi) ? (float)z : 0.0f;
  }
}

template <int DT>
static int launch_bwd(const void* x, const void* gy, void* gx, float* g_delta, float* g_zf, uint64_t n,
                      const tq_quantizer& q, float* ws, size_t ws_bytes, hipStream_t st) {
  constexpr int V = Store<DT>::kVec;
  const bool vector_params = g_delta != nullptr && q.n_params != 1;
  if (vector_params) {
    // element gradient first (scalar kernel), then one reduction block per (parameter, slice)
    const uint64_t outer = n / (q.n_params * q.inner);
    const unsigned slices = bwd_param_slices(outer, q.inner);
    const size_t need = (size_t)q.n_params * slices * 2 * sizeof(float);
    if (ws == nullptr || ws_bytes < need) return set_error(TQ_EWORKSPACE, "tq_fake_quant_bwd: workspace %zu < %zu bytes", ws_bytes, need);
    if (q.n_params > 0x7fffffffull) return set_error(TQ_EUNSUPPORTED, "tq_fake_quant_bwd: too many parameters");
    const unsigned grid1 = (unsigned)std::min<uint64_t>(std::max<uint64_t>(ceil_div(n, kBlock), 1), kMaxGrid);
    hipLaunchKernelGGL((fq_bwd<DT, false>), dim3(grid1), dim3(kBlock), 0, st, x, gy, gx, ws, n, q);
    if (int e = check_launch("fq_bwd")) return e;
    hipLaunchKernelGGL((fq_bwd_params_k<DT>), dim3((unsigned)q.n_params, slices), dim3(kBlock), 0, st, x, gy, outer, q, ws);
    if (int e = check_launch("fq_bwd_params_k")) return e;
    hipLaunchKernelGGL(fq_bwd_params_final_k, dim3((unsigned)ceil_div(q.n_params, 256)), dim3(256), 0, st, ws, slices, q, g_delta,
                       g_zf);
    return check_launch("fq_bwd_params_final_k");
  }
  const bool pgrad = g_delta != nullptr;
  const bool vec_ok = q.n_params == 1 && aligned16(x) && aligned16(gy) && aligned16(gx);
  const uint64_t n_vec = n / V;
  constexpr int U = 2;
  unsigned grid;
  if (vec_ok) grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>(ceil_div(n_vec, kBlock * U), 1), pgrad ? 65536 : kMaxTiles);
  else grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>(ceil_div(n, kBlock), 1), kMaxGrid);
  if (pgrad && (ws == nullptr || ws_bytes < (size_t)grid * 2 * sizeof(float)))
    return set_error(TQ_EWORKSPACE, "tq_fake_quant_bwd: workspace %zu < %zu bytes", ws_bytes, (size_t)grid * 2 * sizeof(float));
  const bool nt = (n * elem_size(DT)) >= ((uint64_t)64 << 20);
  if (vec_ok) {
    const auto xv = static_cast<const u32x4*>(x);
    const auto gv = static_cast<const u32x4*>(gy);
    auto ov = static_cast<u32x4*>(gx);
    if (pgrad) { if (nt) hipLaunchKernelGGL((fq_bwd_tensor<DT, true, true, U>), dim3(grid), dim3(kBlock), 0, st, xv, gv, ov, ws, n, q);
                 else    hipLaunchKernelGGL((fq_bwd_tensor<DT, true, false, U>), dim3(grid), dim3(kBlock), 0, st, xv, gv, ov, ws, n, q); }
    else       { if (nt) hipLaunchKernelGGL((fq_bwd_tensor<DT, false, true, U>), dim3(grid), dim3(kBlock), 0, st, xv, gv, ov, ws, n, q);
                 else    hipLaunchKernelGGL((fq_bwd_tensor<DT, false, false, U>), dim3(grid), dim3(kBlock), 0, st, xv, gv, ov, ws, n, q); }
  } else {
    if (pgrad) hipLaunchKernelGGL((fq_bwd<DT, true>), dim3(grid), dim3(kBlock), 0, st, x, gy, gx, ws, n, q);
    else       hipLaunchKernelGGL((fq_bwd<DT, false>), dim3(grid), dim3(kBlock), 0, st, x, gy, gx, ws, n, q);
  }
  if (int e = check_launch("fq_bwd")) return e;
  if (pgrad) {
    hipLaunchKernelGGL(fq_bwd_final, dim3(1), dim3(kBlock), 0, st, ws, grid, q, g_delta, g_zf);
    return check_launch("fq_bwd_final");
  }
  return TQ_OK;
}


// ------------------------------------------------------------------------------ many independent tensors, one launch
// The 102 weight tensors of a BERT-base are quantized once per range state, one launch each when done lazily by the
// layers (reference hijacker.py:52-64: get_params caches them in eval mode).  Independent sites can share a launch: every
// block finds its tensor in a table that travels as a kernel argument (no upload, capturable), then runs the same element
// arithmetic as the single-tensor kernels (IEEE-division form: bit-identical to all of them).  Per-tensor and
// per-row (per-output-channel weight) parameters; rows must be whole 16-byte vectors.
constexpr int kMultiMax = 40;          // items per launch: the argument block stays below the 4 KB kernarg limit
struct FqMultiArgs {
  const void* x[kMultiMax];
  void* y[kMultiMax];
  uint64_t n[kMultiMax];
  tq_quantizer q[kMultiMax];
  uint32_t first_block[kMultiMax + 1];
  uint32_t count;
};

static_assert(sizeof(FqMultiArgs) <= 4096, "the table must fit the kernel-argument segment");

template <int DT>
__global__ __launch_bounds__(kBlock) void fq_multi_k(FqMultiArgs a) {
  constexpr int V = Store<DT>::kVec;
  typedef typename Store<DT>::elem_t E;
  uint32_t t = 0;
  while (t + 1 < a.count && blockIdx.x >= a.first_block[t + 1]) ++t;      // wave-uniform
  const uint32_t b = blockIdx.x - a.first_block[t], nblk = a.first_block[t + 1] - a.first_block[t];
  const tq_quantizer q = a.q[t];
  const u32x4* x = static_cast<const u32x4*>(a.x[t]);
  u32x4* y = static_cast<u32x4*>(a.y[t]);
  const uint64_t n = a.n[t], n_vec = n / V;
  if (q.n_params == 1) {
    const QP p = make_qp(q, 0);
    for (uint64_t i = (uint64_t)b * kBlock + threadIdx.x; i < n_vec; i += (uint64_t)nblk * kBlock)
      y[i] = fq_vec<DT, false>(x[i], p, nullptr, TQ_IDX_NONE, 0);
    if (b == 0 && n_vec * V + threadIdx.x < n) {
      const uint64_t k = n_vec * V + threadIdx.x;
      Store<DT>::store1(reinterpret_cast<E*>(y) + k, q_dequant(q_index(Store<DT>::load1(reinterpret_cast<const E*>(x) + k), p), p));
    }
  } else {
    const uint64_t vec_per_row = q.inner / V;                              // host: inner % V == 0, n % (n_params * inner) == 0
    for (uint64_t i = (uint64_t)b * kBlock + threadIdx.x; i < n_vec; i += (uint64_t)nblk * kBlock) {
      const QP p = make_qp(q, (i / vec_per_row) % q.n_params);
      y[i] = fq_vec<DT, false>(x[i], p, nullptr, TQ_IDX_NONE, 0);
    }
  }
}

template <int DT>
static int launch_fq_multi(const tq_fq_item* items, uint32_t n_items, hipStream_t st) {
  constexpr int V = Store<DT>::kVec;
  for (uint32_t i0 = 0; i0 < n_items; i0 += kMultiMax) {
    FqMultiArgs a{};
    uint32_t blocks = 0;
    for (uint32_t i = i0; i < n_items && i < i0 + kMultiMax; ++i) {
      const tq_fq_item& it = items[i];
      if (it.n == 0) continue;
      const uint32_t s = a.count++;
      a.x[s] = it.x; a.y[s] = it.y; a.n[s] = it.n; a.q[s] = it.q;
      a.first_block[s] = blocks;
      blocks += (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(it.n / V, (uint64_t)kBlock * 4), 1), 2048);
    }
    if (a.count == 0) continue;
    a.first_block[a.count] = blocks;
    hipLaunchKernelGGL((fq_multi_k<DT>), dim3(blocks), dim3(kBlock), 0, st, a);
    if (int e = check_launch("fq_multi_k")) return e;
  }
  return TQ_OK;
}

}  // namespace tq

using namespace tq;

extern "C" int tq_fake_quant_fwd(const void* x, void* y, void* idx, int idx_dtype, uint64_t n, int dtype,
                                 const tq_quantizer* q, tq_stream_t stream) {
  if (n == 0) return TQ_OK;   // empty tensors are legal (torch semantics) and carry NULL pointers
  TQ_REQUIRE(x != nullptr, "tq_fake_quant_fwd: x is NULL");
  TQ_REQUIRE(y != nullptr || (idx != nullptr && idx_dtype != TQ_IDX_NONE), "tq_fake_quant_fwd: no output requested");
  TQ_REQUIRE(dtype == TQ_F32 || dtype == TQ_BF16 || dtype == TQ_F16, "tq_fake_quant_fwd: bad dtype %d", dtype);
  TQ_REQUIRE(idx_dtype >= TQ_IDX_NONE && idx_dtype <= TQ_IDX_I8_M128, "tq_fake_quant_fwd: bad idx_dtype %d", idx_dtype);
  if (int e = check_quantizer(q, n, "tq_fake_quant_fwd")) return e;
  if (n == 0) return TQ_OK;
  const bool has_idx = idx != nullptr && idx_dtype != TQ_IDX_NONE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case TQ_F32: return has_idx ? launch_fq<TQ_F32, true>(x, y, idx, idx_dtype, n, *q, st) : launch_fq<TQ_F32, false>(x, y, idx, idx_dtype, n, *q, st);
    case TQ_BF16: return has_idx ? launch_fq<TQ_BF16, true>(x, y, idx, idx_dtype, n, *q, st) : launch_fq<TQ_BF16, false>(x, y, idx, idx_dtype, n, *q, st);
    default: return has_idx ? launch_fq<TQ_F16, true>(x, y, idx, idx_dtype, n, *q, st) : launch_fq<TQ_F16, false>(x, y, idx, idx_dtype, n, *q, st);
  }
}

extern "C" int tq_fake_quant_multi_fwd(const tq_fq_item* items, uint32_t n_items, int dtype, tq_stream_t stream) {
  if (n_items == 0) return TQ_OK;
  TQ_REQUIRE(items != nullptr, "tq_fake_quant_multi_fwd: items is NULL");
  TQ_REQUIRE(dtype == TQ_F32 || dtype == TQ_BF16 || dtype == TQ_F16, "tq_fake_quant_multi_fwd: bad dtype %d", dtype);
  const uint64_t V = dtype == TQ_F32 ? 4 : 8;
  for (uint32_t i = 0; i < n_items; ++i) {
    const tq_fq_item& it = items[i];
    if (it.n == 0) continue;
    TQ_REQUIRE(it.x != nullptr && it.y != nullptr, "tq_fake_quant_multi_fwd: item %u has a NULL tensor", i);
    TQ_REQUIRE(aligned16(it.x) && aligned16(it.y), "tq_fake_quant_multi_fwd: item %u: 16-byte alignment required", i);
    if (int e = check_quantizer(&it.q, it.n, "tq_fake_quant_multi_fwd")) return e;
    if (it.q.n_params > 1 && it.q.inner % V != 0)
      return set_error(TQ_EUNSUPPORTED, "tq_fake_quant_multi_fwd: item %u: rows of %llu elements are not whole 16-byte vectors "
                       "(use tq_fake_quant_fwd)", i, (unsigned long long)it.q.inner);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case TQ_F32: return launch_fq_multi<TQ_F32>(items, n_items, st);
    case TQ_BF16: return launch_fq_multi<TQ_BF16>(items, n_items, st);
    default: return launch_fq_multi<TQ_F16>(items, n_items, st);
  }
}

template <int DT>
static int launch_affine(const void* x, const float* w, const float* b, void* y, int8_t* y_idx, uint64_t n, uint64_t d,
                         const tq_quantizer& q, hipStream_t st) {
  constexpr int V = Store<DT>::kVec;
  const uint64_t n_vec = n / V;
  const bool nt = (n * elem_size(DT)) >= ((uint64_t)64 << 20);
  const bool big = n_vec >= (uint64_t)kBlock * 4 * 2048;
  const size_t lds = d * 2 * sizeof(float);
  const auto xv = static_cast<const u32x4*>(x);
  auto yv = static_cast<u32x4*>(y);
#define TQ_LAUNCH_AFF(NTV, UV) hipLaunchKernelGGL((fq_affine<DT, NTV, UV>), dim3((unsigned)std::max<uint64_t>(ceil_div(n_vec, kBlock * UV), 1)), dim3(kBlock), lds, st, xv, w, b, yv, y_idx, n, (uint32_t)d, q)
  if (big) { if (nt) TQ_LAUNCH_AFF(true, 4); else TQ_LAUNCH_AFF(false, 4); }
  else     { if (nt) TQ_LAUNCH_AFF(true, 1); else TQ_LAUNCH_AFF(false, 1); }
#undef TQ_LAUNCH_AFF
  return check_launch("fq_affine");
}

extern "C" int tq_affine_fake_quant_fwd(const void* x, const float* w, const float* b, void* y, int8_t* y_idx, uint64_t n,
                                        uint64_t d, int dtype, const tq_quantizer* q, tq_stream_t stream) {
  if (n == 0) return TQ_OK;
  TQ_REQUIRE(x && w && b && y, "tq_affine_fake_quant_fwd: NULL pointer");
  TQ_REQUIRE(dtype == TQ_F32 || dtype == TQ_BF16 || dtype == TQ_F16, "tq_affine_fake_quant_fwd: bad dtype %d", dtype);
  if (int e = check_quantizer(q, n, "tq_affine_fake_quant_fwd")) return e;
  TQ_REQUIRE(q->n_params == 1, "tq_affine_fake_quant_fwd: per-tensor output quantizer only");
  const uint64_t V = dtype == TQ_F32 ? 4 : 8;
  TQ_REQUIRE(d >= V && d % V == 0 && d <= 8192 && n % d == 0, "tq_affine_fake_quant_fwd: d=%llu unsupported", (unsigned long long)d);
  TQ_REQUIRE(aligned16(x) && aligned16(y), "tq_affine_fake_quant_fwd: x / y must be 16-byte aligned");
  TQ_REQUIRE(y_idx == nullptr || (!q->symmetric && q->n_bits <= 8 && (reinterpret_cast<uintptr_t>(y_idx) & 7u) == 0),
             "tq_affine_fake_quant_fwd: y_idx needs an asymmetric <= 8-bit quantizer and 8-byte alignment");
  TQ_REQUIRE(n / V / (kBlock) < (1ull << 31), "tq_affine_fake_quant_fwd: tensor too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case TQ_F32: return launch_affine<TQ_F32>(x, w, b, y, y_idx, n, d, *q, st);
    case TQ_BF16: return launch_affine<TQ_BF16>(x, w, b, y, y_idx, n, d, *q, st);
    default: return launch_affine<TQ_F16>(x, w, b, y, y_idx, n, d, *q, st);
  }
}

extern "C" size_t tq_fake_quant_bwd_workspace_bytes(uint64_t n) { return (size_t)65536 * 2 * sizeof(float); }

extern "C" size_t tq_fake_quant_bwd_params_workspace_bytes(uint64_t n, uint64_t n_params, uint64_t inner) {
  if (n_params <= 1) return tq_fake_quant_bwd_workspace_bytes(n);
  const uint64_t outer = inner ? n / (n_params * inner) : 0;
  return (size_t)n_params * bwd_param_slices(outer, inner) * 2 * sizeof(float);
}

extern "C" int tq_fake_quant_bwd(const void* x, const void* grad_y, void* grad_x, float* grad_delta,
                                 float* grad_zero_float, uint64_t n, int dtype, const tq_quantizer* q,
                                 void* workspace, size_t workspace_bytes, tq_stream_t stream) {
  if (n == 0) return TQ_OK;
  TQ_REQUIRE(x && grad_y && grad_x, "tq_fake_quant_bwd: NULL tensor");
  TQ_REQUIRE(dtype == TQ_F32 || dtype == TQ_BF16 || dtype == TQ_F16, "tq_fake_quant_bwd: bad dtype %d", dtype);
  if (int e = check_quantizer(q, n, "tq_fake_quant_bwd")) return e;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* ws = static_cast<float*>(workspace);
  switch (dtype) {
    case TQ_F32: return launch_bwd<TQ_F32>(x, grad_y, grad_x, grad_delta, grad_zero_float, n, *q, ws, workspace_bytes, st);
    case TQ_BF16: return launch_bwd<TQ_BF16>(x, grad_y, grad_x, grad_delta, grad_zero_float, n, *q, ws, workspace_bytes, st);
    default: return launch_bwd<TQ_F16>(x, grad_y, grad_x, grad_delta, grad_zero_float, n, *q, ws, workspace_bytes, st);
  }
}
