// (f2) Fused residual-add -> quantize -> LayerNorm -> quantize for gfx950.
//
// The tail of BertSelfOutput / BertOutput in the reference (models/quantized_bert.py:238-248,
// 264-280 with quantization/hijacker.py:98-116 and autoquant_utils.py:55-66) is five separate
// tensor sweeps after the GEMM:
//     t = Q1(dense_out)        activation quantizer of the QuantLinear
//     s = t + residual
//     u = Q2(s)                res_act_quantizer
//     v = layer_norm(u; Q(w), Q(b), eps)
//     y = Q3(v)                activation quantizer of the QuantLayerNorm
// = 5 reads + 4 writes of [B*T, d] plus ~20 launches.  With fixed ranges the chain only needs the
// row it is working on, so it collapses to 2 reads + 1 write (6 B/elem bf16, 12 B/elem fp32):
// LPR lanes (16/32/64) own one row, keep it in registers (d/LPR values per lane), do the two
// row reductions (mean, centred variance) with __shfl_xor inside the lane group, and stream out.
// Any of the three quantizers may be absent (NULL).
#include <algorithm>

#include "tq_device.h"
#include "tq_host.h"

namespace tq {

struct FusedQ {
  QP p;
  int on;
  float rcp;     // guarded_rcp(p.scale): three quantizers per element make these kernels VALU-bound with the division
};

__device__ __forceinline__ FusedQ make_fq(const QP& p, int on) {
  FusedQ f = {QP{1.f, 0.f, 0.f, 0.f}, on, 1.f};
  if (on) { f.p = p; f.rcp = guarded_rcp(f.p.scale); }
  return f;
}
__device__ __forceinline__ FusedQ make_fq(const tq_quantizer& q, int on) { return make_fq(on ? make_qp(q, 0) : QP{1.f, 0.f, 0.f, 0.f}, on); }

// x_int of v under q (bit-identical to q_index: rne_quot1 falls back to the division near ties)
__device__ __forceinline__ float index_q(float v, const FusedQ& q) {
  return clamp_nanprop(rne_quot1(v, q.p.scale, q.rcp) + q.p.zp, q.p.lo, q.p.hi);
}

__device__ __forceinline__ float apply_q(float v, const FusedQ& q) {
  return q.on ? q_dequant(index_q(v, q), q.p) : v;
}

// Sum over the LPR lanes that own one row (every lane gets the total).  The first four butterfly steps run on the
// VALU's data-parallel-primitive path (quad permutes, half-row / row mirrors: one v_add_f32_dpp each, no index
// arithmetic, no LDS crossbar round trip); lanes 16 apart through ds_swizzle, 32 apart through ds_bpermute.
// (__shfl_xor costs 4 VALU instructions of index math + a dependent ds_bpermute per step: 10 steps per LayerNorm row
// were ~110 instructions and the longest dependency chain of the kernel.)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
  static_assert(LPR >= 4 && LPR <= 64 && (LPR & (LPR - 1)) == 0, "LPR");
  v += dpp_f32<0xB1>(v);                        // quad_perm [1, 0, 3, 2]
  v += dpp_f32<0x4E>(v);                        // quad_perm [2, 3, 0, 1]
  if (LPR >= 8) v += dpp_f32<0x141>(v);         // row_half_mirror: the other quad of the 8 (all its lanes agree by now)
  if (LPR >= 16) v += dpp_f32<0x140>(v);        // row_mirror: the other half of the 16
  if (LPR >= 32) v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));   // lane ^ 16
  if (LPR >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

template <int LPR>
__device__ __forceinline__ float group_max(float v) {      // fmaxf butterfly on the same paths as group_sum
  v = max_raw(v, dpp_f32<0xB1>(v));
  v = max_raw(v, dpp_f32<0x4E>(v));
  if (LPR >= 8) v = max_raw(v, dpp_f32<0x141>(v));
  if (LPR >= 16) v = max_raw(v, dpp_f32<0x140>(v));
  if (LPR >= 32) v = max_raw(v, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F)));
  if (LPR >= 64) v = max_raw(v, __shfl_xor(v, 32, 64));
  return v;
}

struct FusedF {
  QF f;
  int on;
};
__device__ __forceinline__ FusedF make_ff(const QP& p, int on) {
  FusedF r;
  r.on = on;
  r.f = make_qf(on ? p : QP{1.f, 0.f, 0.f, 1.f});
  return r;
}
__device__ __forceinline__ f32x2 apply_f2(f32x2 v, const FusedF& q) { return q.on ? qf_fake_quant2(v, q.f) : v; }

// NV = 16-byte vectors per lane per row (d == LPR * NV * V).  All element math runs on register pairs with the exact
// branch-free quantizer of tq_device.h (QF): three quantizers per element made the scalar version VALU- / latency-bound
// on bf16 rows (3.1-3.3 TB/s).  NaN: an unordered input pair poisons its element before the statistics (so a LayerNorm
// row turns NaN as a whole, like the reference) and a NaN pre-quantizer value is passed through at the end.  FAST = false
// is the division path (scales outside [2^-100, 2^100], grids of 2^22+ steps); IDX also emits int8(index - 128).
// BERT's embedding block through the same body (EMB): the "dense output" row is word[word_ids[row]] + type[type_ids[row]]
// (one fp32 addition, like the reference's `inputs_embeds + token_type_embeddings`, models/quantized_bert.py:95-111), the
// "residual" row is pos[pos_ids[row]].  An id outside its table (torch's CPU F.embedding raises IndexError there; a GPU
// look-up would read out of bounds) reads the clamped row for memory safety, turns the WHOLE output row NaN (a corrupt
// batch or a vocabulary mismatch must not produce plausible activations) and raises `*bad` (optional; host-visible
// memory: the Python side turns it into the IndexError of the layered CPU route, quantization/_hip.py).
struct EmbArgs {
  const int64_t *a_rows, *a2_rows, *r_rows;
  const u32x4* a2;
  uint64_t a_n, a2_n, r_n;        // rows of the three tables
  int32_t* bad;                   // set to 1 by any row with an out-of-range id (nullptr: not reported)
};

// One row's 16-byte vectors of this lane (dense output, residual, and the token-type row in EMB mode) -> registers.
// Streaming hints only for tensors that cannot stay in L2 / MALL anyway: in a model forward the inputs were just written by
// the GEMM and the output is read by the next layer.  ONE uniform branch per group of loads.
template <int DT, int LPR, int NV, bool EMB>
__device__ __forceinline__ void ln_load_row(const u32x4* __restrict__ a, const u32x4* __restrict__ r, const EmbArgs& emb, int nt,
                                            int lane, uint64_t row, u32x4 (&pa)[NV], u32x4 (&pr)[NV], u32x4 (&pa2)[EMB ? NV : 1]) {
  constexpr int V = Store<DT>::kVec;
  constexpr uint32_t d = LPR * NV * V;
  if (EMB) {
    bool bad = false;
    auto pick = [&bad](const int64_t* ids, uint64_t row_, uint64_t n) {
      const int64_t id = ids[row_];
      const bool out = id < 0 || (uint64_t)id >= n;
      bad = bad || out;
      return out ? (uint64_t)0 : (uint64_t)id;
    };
    const uint64_t oa = pick(emb.a_rows, row, emb.a_n) * (d / V) + lane;
    const uint64_t o2 = pick(emb.a2_rows, row, emb.a2_n) * (d / V) + lane;
    const uint64_t orr = pick(emb.r_rows, row, emb.r_n) * (d / V) + lane;
#pragma unroll
    for (int v = 0; v < NV; ++v) { pa[v] = a[oa + v * LPR]; pa2[v] = emb.a2[o2 + v * LPR]; pr[v] = r[orr + v * LPR]; }
    if (bad) {                       // row-uniform and rare: poison the word row (NaN rule of the body: the row turns NaN)
      const u32x4 nan4 = {0x7fc00000u, 0x7fc00000u, 0x7fc00000u, 0x7fc00000u};
#pragma unroll
      for (int v = 0; v < NV; ++v) pa[v] = nan4;
      if (lane == 0 && emb.bad) __hip_atomic_store(emb.bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  const uint64_t o = row * (d / V) + lane;
  if (nt) {
#pragma unroll
    for (int v = 0; v < NV; ++v) { pa[v] = ld_stream(a + o + v * LPR); pr[v] = ld_stream(r + o + v * LPR); }
  } else {
#pragma unroll
    for (int v = 0; v < NV; ++v) { pa[v] = a[o + v * LPR]; pr[v] = r[o + v * LPR]; }
  }
}

// The three quantizers as the kernel entry derived them (once, from one batch of loads: tq_device.h load_qraw).
struct TailQ {
  QP p1, p2, p3;
};

template <int DT, int LPR, int NV, bool IDX, bool FAST, bool ALLON, bool AFF, bool EMB>
__device__ __forceinline__ void res_ln_body(const u32x4* __restrict__ a, const u32x4* __restrict__ r,
                                            u32x4* __restrict__ y, int8_t* __restrict__ y_idx, uint64_t rows,
                                            const float* s_w, const float* s_b,
                                            float ln_eps, const TailQ& tq, int on1_, int on2_, int on3_, int nt, uint32_t iters,
                                            const EmbArgs& emb, u32x4 (&va)[NV], u32x4 (&vr)[NV], u32x4 (&va2)[EMB ? NV : 1]) {
  // AFF (NoNorm: affine map only, no statistics) is a compile-time property: as a run-time flag every NaN rule of BOTH
  // variants was evaluated per element and selected (4-5 v_cndmask / v_cmp per element of ~29 VALU instructions on the
  // bf16 LayerNorm rows, found in the ISA in round 4)
  constexpr bool affine_only = AFF;
  // ALLON: the three quantizers are present (the configuration the tails exist for) -> compile-time flags, no uniform
  // branch around every quantizer application (those branches split the row into ~40 basic blocks and serialised it)
  const int on1 = ALLON ? 1 : on1_, on2 = ALLON ? 1 : on2_, on3 = ALLON ? 1 : on3_;
  constexpr int V = Store<DT>::kVec;
  constexpr int H = V / 2;                          // register pairs per 16-byte vector
  constexpr int RPB = kBlock / LPR;                 // rows per block iteration
  constexpr uint32_t d = LPR * NV * V;
  const FusedF f1 = make_ff(tq.p1, on1), f2 = make_ff(tq.p2, on2), f3 = make_ff(tq.p3, on3);
  const FusedQ g1 = make_fq(tq.p1, on1), g2 = make_fq(tq.p2, on2), g3 = make_fq(tq.p3, on3);     // division path (FAST = false)
  const int lane = threadIdx.x % LPR;
  const int sub = threadIdx.x / LPR;
  const float inv_d = 1.0f / (float)d;

  // A block owns `iters` consecutive groups of RPB rows (one-shot tiles in row order: the resident blocks sweep one
  // contiguous window of HBM); the loads of group i + 1 are issued before group i is computed, and the affine
  // parameters / quantizer constants are set up once per block instead of once per RPB rows.  Measured on
  // [131072, 768] (tools/tuning/tail_sweep.py): bf16 LayerNorm 3.6 / 4.6 / 4.9 / 4.9 TB/s for iters = 1 / 2 / 4 / 8,
  // fp32 5.4 / 5.8 / 5.4 / 5.2 -> 8 for 2-byte storage, 2 for fp32 (launch_res_ln).  The first group's loads were issued by
  // the kernel entry, ahead of everything else.
  const uint64_t row0 = (uint64_t)blockIdx.x * RPB * iters + sub;
  u32x4 na[NV], nr[NV];
  u32x4 na2[EMB ? NV : 1];
  auto load_row = [&](uint64_t row, u32x4 (&pa)[NV], u32x4 (&pr)[NV], u32x4 (&pa2)[EMB ? NV : 1]) {
    ln_load_row<DT, LPR, NV, EMB>(a, r, emb, nt, lane, row, pa, pr, pa2);
  };
  for (uint32_t it = 0; it < iters; ++it) {
    const uint64_t row = row0 + (uint64_t)it * RPB;
    if (row >= rows) break;
    const uint64_t base = row * (d / V);
    const uint64_t nrow = row + RPB;
    if (it + 1 < iters && nrow < rows) load_row(nrow, na, nr, na2);
    f32x2 u[NV][H];
    f32x2 s2 = {0.f, 0.f}, s2b = {0.f, 0.f};      // two chains for the row sum (LayerNorm only)
    bool lane_nan = false;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      float fa[V], fr[V];
      Store<DT>::unpack(va[v], fa);
      Store<DT>::unpack(vr[v], fr);
      if (EMB) {                                    // word row + token-type row: ONE fp32 addition in front of Q1
        float f2[V];
        Store<DT>::unpack(va2[v], f2);
#pragma unroll
        for (int k = 0; k < V; ++k) fa[k] = fa[k] + f2[k];
      }
      if (FAST) {
        // the H pairs of this 16-byte vector move through the stages side by side (qf_*_n)
        f32x2 t[H];
#pragma unroll
        for (int j = 0; j < H; ++j) t[j] = f32x2{fa[2 * j], fa[2 * j + 1]};
        // ALLON: Q3 follows, so a -0 out of Q1 / Q2 cannot reach y (Q1(a) + r, the statistics and the affine map treat
        // -0 like +0 unless every term is a zero, and Q3 normalises a surviving zero): skip their "+ 0"
        if (ALLON) qf_fake_quant2_n_signed_zero<H>(t, f1.f);
        else if (f1.on) qf_fake_quant2_n<H>(t, f1.f);
#pragma unroll
        for (int j = 0; j < H; ++j) t[j] = t[j] + f32x2{fr[2 * j], fr[2 * j + 1]};
        if (ALLON) qf_fake_quant2_n_signed_zero<H>(t, f2.f);
        else if (f2.on) qf_fake_quant2_n<H>(t, f2.f);
        if (affine_only) {        // NoNorm: a NaN input is an element-local NaN output
#pragma unroll
          for (int j = 0; j < H; ++j) {
            t[j].x = __builtin_isunordered(fa[2 * j], fr[2 * j]) ? __builtin_nanf("") : t[j].x;
            t[j].y = __builtin_isunordered(fa[2 * j + 1], fr[2 * j + 1]) ? __builtin_nanf("") : t[j].y;
          }
        } else {                  // LayerNorm: any NaN turns the WHOLE row NaN -> one compare per element, masks OR-ed
#pragma unroll
          for (int j = 0; j < H; ++j)
            lane_nan = lane_nan || __builtin_isunordered(fa[2 * j], fr[2 * j]) ||
                       __builtin_isunordered(fa[2 * j + 1], fr[2 * j + 1]);
        }
#pragma unroll
        for (int j = 0; j < H; ++j) u[v][j] = t[j];
      } else {
#pragma unroll
        for (int j = 0; j < H; ++j)
          u[v][j] = f32x2{apply_q(apply_q(fa[2 * j], g1) + fr[2 * j], g2),
                          apply_q(apply_q(fa[2 * j + 1], g1) + fr[2 * j + 1], g2)};
      }
      if (!affine_only) {
#pragma unroll
        for (int j = 0; j < H; ++j) {
          if (j & 1) s2b = s2b + u[v][j];
          else s2 = s2 + u[v][j];
        }
      }
    }
    s2 = s2 + s2b;
    // MobileBERT's NoNorm (models/quantized_mobilebert.py:58-72) is the affine part alone: u * w + b.  With
    // mean = 0 and rstd = 1 the expression below evaluates exactly that ((u - 0) * 1 is exact).
    float mean = 0.0f, rstd = 1.0f;
    bool row_nan = false;                 // LayerNorm: the whole row is NaN (wave-uniform branch at the store, rare)
    if (!affine_only) {
      mean = group_sum<LPR>(s2.x + s2.y) * inv_d;
      const f32x2 m2 = {mean, mean};
      // centred sum of squares: two independent packed accumulators fed by fma (a single `ss2 = ss2 + c * c` chain was
      // 2 packed ops per pair plus a wait state after each: 46 s_nop per 24-element lane row in the round-3 ISA).  The
      // LayerNorm statistics carry a tolerance contract (tests/test_fused_ln.py); NoNorm has no statistics.
      f32x2 ssa = {0.f, 0.f}, ssb = {0.f, 0.f};
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int j = 0; j < H; ++j) {
          const f32x2 c = u[v][j] - m2;
          if ((v * H + j) & 1) ssb = __builtin_elementwise_fma(c, c, ssb);
          else ssa = __builtin_elementwise_fma(c, c, ssa);
        }
      const f32x2 ss2 = ssa + ssb;
      rstd = 1.0f / sqrtf(group_sum<LPR>(ss2.x + ss2.y) * inv_d + ln_eps);
      if (FAST) {
        // rows with a NaN input: the statistics are NaN upstream, hence every output of the row
        const uint64_t m = __ballot(lane_nan);
        const uint64_t grp = LPR == 64 ? ~0ull : (((1ull << (LPR % 64)) - 1) << ((threadIdx.x & 63) / LPR * LPR));
        if (m & grp) { mean = __builtin_nanf(""); row_nan = true; }
      }
    }
    const f32x2 m2 = {mean, mean}, r2 = {rstd, rstd};
    u32x4 packed[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      float o[V];
      struct alignas(V) { int8_t e[V]; } oi;
      f32x2 t[H];
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const uint32_t c = (v * LPR + lane) * V + 2 * j;
        const f32x2 wv = *reinterpret_cast<const f32x2*>(s_w + c), bv = *reinterpret_cast<const f32x2*>(s_b + c);
        t[j] = (u[v][j] - m2) * r2 * wv + bv;
      }
      if (f3.on) {
        if (FAST) {
          f32x2 h[H];
          qf_round2_n<H>(t, f3.f, h);
#pragma unroll
          for (int j = 0; j < H; ++j) {
            if (IDX) {
              oi.e[2 * j] = (int8_t)((int)(h[j].x + f3.f.zp) - 128);
              oi.e[2 * j + 1] = (int8_t)((int)(h[j].y + f3.f.zp) - 128);
            }
            const f32x2 yq = qf_dequant2(h[j], f3.f);                     // -0 -> +0 like (x_int - zp), one fma
            if (affine_only) {            // NoNorm: element-local NaN passes through
              t[j].x = (t[j].x != t[j].x) ? t[j].x : yq.x;
              t[j].y = (t[j].y != t[j].y) ? t[j].y : yq.y;
            } else {
              t[j] = yq;                  // LayerNorm: NaN rows are patched below
            }
          }
          // (LayerNorm rows with a NaN input are overwritten AFTER the stores below, in a branch that is practically never
          // taken: patching t[] here became one select per element of every row)
        } else {
#pragma unroll
          for (int j = 0; j < H; ++j) {
            const float x0 = index_q(t[j].x, g3), x1 = index_q(t[j].y, g3);
            if (IDX) {
              oi.e[2 * j] = (int8_t)((int)x0 - 128);
              oi.e[2 * j + 1] = (int8_t)((int)x1 - 128);
            }
            t[j] = f32x2{q_dequant(x0, g3.p), q_dequant(x1, g3.p)};
          }
        }
      }
#pragma unroll
      for (int j = 0; j < H; ++j) { o[2 * j] = t[j].x; o[2 * j + 1] = t[j].y; }
      packed[v] = Store<DT>::pack(o);
      if (IDX) *reinterpret_cast<decltype(oi)*>(y_idx + (base + v * LPR + lane) * V) = oi;
    }
    if (nt) {
#pragma unroll
      for (int v = 0; v < NV; ++v) st_stream(y + base + v * LPR + lane, packed[v]);
    } else {
#pragma unroll
      for (int v = 0; v < NV; ++v) y[base + v * LPR + lane] = packed[v];
    }
    if (FAST && !affine_only && f3.on && __ballot(row_nan)) {          // rare: a LayerNorm row with a NaN input is NaN as a whole
      if (row_nan) {
        float nanv[V];
#pragma unroll
        for (int e = 0; e < V; ++e) nanv[e] = __builtin_nanf("");
        const u32x4 pn = Store<DT>::pack(nanv);
#pragma unroll
        for (int v = 0; v < NV; ++v) y[base + v * LPR + lane] = pn;
      }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) { va[v] = na[v]; vr[v] = nr[v]; }
    if (EMB) {
#pragma unroll
      for (int v = 0; v < NV; ++v) va2[v] = na2[v];
    }
  }
}

template <int DT, int LPR, int NV, bool IDX, bool AFF, bool EMB = false>
__global__ __launch_bounds__(kBlock) void res_ln_quant_k(const u32x4* __restrict__ a, const u32x4* __restrict__ r,
                                                         u32x4* __restrict__ y, int8_t* __restrict__ y_idx, uint64_t rows,
                                                         const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                         float ln_eps, tq_quantizer q1, tq_quantizer q2, tq_quantizer q3,
                                                         int on1, int on2, int on3, int nt, uint32_t iters, EmbArgs emb) {
  // At a model's inference shapes ([1024, 768]: 4 rows per block) this kernel IS its prologue, so everything it reads is
  // requested before anything is waited for: the first row of every lane group, the affine parameters (staged in LDS,
  // read as 8-byte pairs where they are used: keeping this lane's 2 x d / LPR values in registers cost 48 VGPRs on bf16
  // rows and with them half the occupancy), and the raw buffers of the three quantizers -- one memory round trip where
  // the chain rows <- affine staging <- code-path choice <- quantizer by quantizer took ~13 (7.6 -> 4.x us per launch).
  constexpr int V = Store<DT>::kVec;
  constexpr int RPB = kBlock / LPR;
  constexpr uint32_t d = LPR * NV * V;
  constexpr int NA = (d + kBlock - 1) / kBlock;
  __shared__ __attribute__((aligned(16))) float s_w[d], s_b[d];
  u32x4 va[NV], vr[NV];
  u32x4 va2[EMB ? NV : 1];
  const uint64_t row0 = (uint64_t)blockIdx.x * RPB * iters + threadIdx.x / LPR;
  if (row0 < rows) ln_load_row<DT, LPR, NV, EMB>(a, r, emb, nt, threadIdx.x % LPR, row0, va, vr, va2);
  float aw[NA], ab[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const uint32_t c = threadIdx.x + i * kBlock;
    const uint32_t cc = c < d ? c : d - 1;
    aw[i] = ln_w[cc];
    ab[i] = ln_b[cc];
  }
  QRaw w1 = load_qraw(q1, 0, ln_w), w2 = load_qraw(q2, 0, ln_w), w3 = load_qraw(q3, 0, ln_w);
  qraw_arrived(w1); qraw_arrived(w2); qraw_arrived(w3);
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const uint32_t c = threadIdx.x + i * kBlock;
    if (c < d) { s_w[c] = aw[i]; s_b[c] = ab[i]; }
  }
  __syncthreads();
  const QP none = {1.f, 0.f, 0.f, 0.f};
  const TailQ tq = {on1 ? qp_from_raw(q1, w1) : none, on2 ? qp_from_raw(q2, w2) : none, on3 ? qp_from_raw(q3, w3) : none};
  // wave-uniform: every enabled quantizer admits the branch-free exact path (tq_device.h, QF)
  bool fast = true;
  if (on1) fast = fast && make_qf(tq.p1).ok;
  if (on2) fast = fast && make_qf(tq.p2).ok;
  if (on3) fast = fast && make_qf(tq.p3).ok;
  if (fast && on1 && on2 && on3)
    res_ln_body<DT, LPR, NV, IDX, true, true, AFF, EMB>(a, r, y, y_idx, rows, s_w, s_b, ln_eps, tq, 1, 1, 1, nt, iters, emb, va, vr, va2);
  else if (fast)
    res_ln_body<DT, LPR, NV, IDX, true, false, AFF, EMB>(a, r, y, y_idx, rows, s_w, s_b, ln_eps, tq, on1, on2, on3, nt, iters, emb, va, vr, va2);
  else
    res_ln_body<DT, LPR, NV, IDX, false, false, AFF, EMB>(a, r, y, y_idx, rows, s_w, s_b, ln_eps, tq, on1, on2, on3, nt, iters, emb, va, vr, va2);
}

template <int DT>
static int launch_res_ln(const void* a, const void* r, void* y, int8_t* y_idx, uint64_t rows, uint64_t d, const float* w, const float* b,
                         float eps, const tq_quantizer* q1, const tq_quantizer* q2, const tq_quantizer* q3, int affine_only,
                         hipStream_t st, const EmbArgs* emb_in = nullptr) {
  const EmbArgs emb = emb_in ? *emb_in : EmbArgs{};
  constexpr int V = Store<DT>::kVec;
  const tq_quantizer none{};
  const tq_quantizer &c1 = q1 ? *q1 : none, &c2 = q2 ? *q2 : none, &c3 = q3 ? *q3 : none;
  const auto av = static_cast<const u32x4*>(a);
  const auto rv = static_cast<const u32x4*>(r);
  auto yv = static_cast<u32x4*>(y);
  const uint64_t vpr = d / V;
  const int nt = rows * d * elem_size(DT) >= (64ull << 20);
#define TQ_LN_GO(LPR, NV, IDXV, AFFV)                                                                           \
  hipLaunchKernelGGL((res_ln_quant_k<DT, LPR, NV, IDXV, AFFV>), dim3(grid), dim3(kBlock), 0, st, av, rv, yv, y_idx, rows, w, b, \
                     eps, c1, c2, c3, q1 != nullptr, q2 != nullptr, q3 != nullptr, nt, iters, emb)
#define TQ_LN(LPR, NV)                                                                                          \
  if (vpr == (uint64_t)(LPR) * (NV)) {                                                                          \
    const unsigned rpb = kBlock / (LPR);                                                                        \
    const uint32_t iters = (uint32_t)std::max<int>(1, std::min<uint64_t>(tuning("TQ_TAIL_ITERS", DT == TQ_F32 ? 2 : 8), ceil_div(rows, (uint64_t)rpb * 2048))); \
    const unsigned grid = (unsigned)std::max<uint64_t>(ceil_div(rows, (uint64_t)rpb * iters), 1);               \
    if (DT == TQ_F32 && emb_in != nullptr) {                                                                    \
      if (y_idx != nullptr) hipLaunchKernelGGL((res_ln_quant_k<TQ_F32, LPR, NV, true, false, true>), dim3(grid), dim3(kBlock), 0, st, av, rv, yv, y_idx, rows, w, b, \
                                               eps, c1, c2, c3, q1 != nullptr, q2 != nullptr, q3 != nullptr, 0, iters, emb); \
      else hipLaunchKernelGGL((res_ln_quant_k<TQ_F32, LPR, NV, false, false, true>), dim3(grid), dim3(kBlock), 0, st, av, rv, yv, y_idx, rows, w, b, \
                              eps, c1, c2, c3, q1 != nullptr, q2 != nullptr, q3 != nullptr, 0, iters, emb);   \
      return check_launch("res_ln_quant_k (embeddings)");                                                       \
    }                                                                                                           \
    if (y_idx != nullptr && affine_only)      TQ_LN_GO(LPR, NV, true, true);                                    \
    else if (y_idx != nullptr)                TQ_LN_GO(LPR, NV, true, false);                                   \
    else if (affine_only)                     TQ_LN_GO(LPR, NV, false, true);                                   \
    else                                      TQ_LN_GO(LPR, NV, false, false);                                  \
    return check_launch("res_ln_quant_k");                                                                      \
  }
  // d (bf16 | fp32): 768 -> 96 | 192 vectors, 3072 -> 384 | 768, 512 -> 64 | 128, 128 -> 16 | 32, 1024 -> 128 | 256
  TQ_LN(32, 3) TQ_LN(64, 3) TQ_LN(64, 6) TQ_LN(64, 12) TQ_LN(64, 1) TQ_LN(64, 2) TQ_LN(64, 4) TQ_LN(16, 1) TQ_LN(32, 1) TQ_LN(64, 8)
#undef TQ_LN
#undef TQ_LN_GO
  return set_error(TQ_EUNSUPPORTED, "tq_residual_layernorm_quant_fwd: row length %llu has no instantiation",
                   (unsigned long long)d);
}

// ---------------------------------------------------------------------------------------------------
// Attention probabilities with fixed ranges (reference models/quantized_bert.py:153-198):
//     p = Q_probs( softmax( Q_scores(scores) / denom + mask , dim=-1 ) )
// = quantizer, division, mask add, softmax, quantizer: five sweeps of [B, H, T, T] in the reference,
// here 1 read + 1 write.  LPR lanes own one row of T = LPR * NV * 4 fp32 values.
// Branch-free exact variant of the element math (both quantizers on, tq_device.h QF): ~30 VALU issue slots per element
// instead of ~75 with four IEEE divisions -- the scalar version was VALU-bound at 40 % of the HBM rate.
//   scores quantizer, probs quantizer: QF pairs (med3 clamp + Markstein quotient);
//   t / denom: Markstein with rd = RN(1 / denom): the operand is a grid value (>= 2^-60 or 0, < 2^82), so the fma
//     residual is exact and the quotient correctly rounded -- the same value v_div_* returns;
//   e / sum: the same with rs = RN(1 / sum), one IEEE division per row.  1 <= sum <= T and e in [0, 1]: the residual
//     is exact for e >= 2^-62; below that the probability is < scale / 4 and the probs quantizer returns index 0
//     whatever the last bit of the quotient (which is why this path needs the probs quantizer).
// NaN: v_med3 does not keep it, so a NaN score poisons the row sum (the reference's row is NaN as a whole) and rows
// with a NaN sum are patched after the last quantizer.
__device__ __forceinline__ bool softmax_fast_ok(const tq_quantizer& q1, const tq_quantizer& q2, int on1, int on2, float denom) {
  if (!on1 || !on2) return false;
  // (make_qp's select form in the softmax chain: with the branch form of round 6 the [256,12,128,128] launch went from 75 to
  // 82-86 us in the kernel table; 86 -> 80 us on one box with the select, profiles/r06/softmax_select_ab.txt)
  const QP p1 = make_qp<true>(q1, 0), p2 = make_qp<true>(q2, 0);
  const float ad = fabsf(denom);
  return make_qf(p1).ok && make_qf(p2).ok && p1.scale >= 0x1p-60f && p1.scale <= 0x1p60f && p2.scale >= 0x1p-60f &&
         p2.scale <= 0x1p60f && ad >= 0x1p-20f && ad <= 0x1p20f;
}

template <int LPR, int NV, int R>
__device__ __forceinline__ void softmax_fast_body(const f32x4* __restrict__ s, f32x4* __restrict__ y, uint64_t rows,
                                                  const float* __restrict__ mask, uint64_t rows_per_mask, float denom,
                                                  const tq_quantizer& q1, const tq_quantizer& q2) {
  constexpr int RPB = kBlock / LPR;
  constexpr uint32_t T4 = LPR * NV;
  constexpr int P = NV * 2;                      // pairs per lane and row
  const QF f1 = make_qf(make_qp<true>(q1, 0)), f2 = make_qf(make_qp<true>(q2, 0));
  const float rdv = 1.0f / denom;
  const f32x2 rd = {rdv, rdv}, nd = {-denom, -denom};
  const int lane = threadIdx.x % LPR, sub = threadIdx.x / LPR;
  for (uint64_t row0 = ((uint64_t)blockIdx.x * RPB + sub) * R; row0 < rows; row0 += (uint64_t)gridDim.x * RPB * R) {
    f32x4 in[R][NV], mk[R][NV];
    const uint64_t m0 = mask ? row0 / rows_per_mask : 0, rem0 = row0 - m0 * rows_per_mask;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint64_t row = row0 + r < rows ? row0 + r : rows - 1;
      const uint64_t mi = rem0 + r < rows_per_mask ? m0 : row / rows_per_mask;
      const f32x4* mrow = mask ? reinterpret_cast<const f32x4*>(mask + mi * (uint64_t)T4 * 4) : nullptr;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        in[r][k] = s[row * T4 + k * LPR + lane];
        mk[r][k] = mrow ? mrow[k * LPR + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    f32x2 t[R][P];
    float mx[R], sum[R];
    bool bad[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      bad[r] = false;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        bad[r] = bad[r] || in[r][k][0] != in[r][k][0] || in[r][k][1] != in[r][k][1] || in[r][k][2] != in[r][k][2] ||
                 in[r][k][3] != in[r][k][3];
        t[r][2 * k] = f32x2{in[r][k][0], in[r][k][1]};
        t[r][2 * k + 1] = f32x2{in[r][k][2], in[r][k][3]};
      }
      qf_fake_quant2_n<P>(t[r], f1);
      f32x2 q0[P], e[P];
#pragma unroll
      for (int i = 0; i < P; ++i) q0[i] = t[r][i] * rd;
#pragma unroll
      for (int i = 0; i < P; ++i) e[i] = __builtin_elementwise_fma(q0[i], nd, t[r][i]);
#pragma unroll
      for (int i = 0; i < P; ++i) t[r][i] = __builtin_elementwise_fma(e[i], rd, q0[i]);
      if (mask) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          t[r][2 * k] = t[r][2 * k] + f32x2{mk[r][k][0], mk[r][k][1]};
          t[r][2 * k + 1] = t[r][2 * k + 1] + f32x2{mk[r][k][2], mk[r][k][3]};
        }
      }
      mx[r] = -__builtin_huge_valf();
#pragma unroll
      for (int i = 0; i < P; ++i) mx[r] = max_raw(mx[r], max_raw(t[r][i].x, t[r][i].y));
    }
#pragma unroll
    for (int r = 0; r < R; ++r) mx[r] = group_max<LPR>(mx[r]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      sum[r] = 0.f;
#pragma unroll
      for (int i = 0; i < P; ++i) {
        t[r][i].x = exp_nonpos_fast(t[r][i].x - mx[r]);      // tolerance contract (see tq_device.h), ~8 slots instead of ~13
        t[r][i].y = exp_nonpos_fast(t[r][i].y - mx[r]);
        sum[r] += t[r][i].x;             // the scalar kernel's (and the oracle's) left-to-right order
        sum[r] += t[r][i].y;
      }
      if (bad[r]) sum[r] = __builtin_nanf("");
    }
#pragma unroll
    for (int r = 0; r < R; ++r) sum[r] = group_sum<LPR>(sum[r]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float rsv = 1.0f / sum[r];
      const f32x2 rs = {rsv, rsv}, ns = {-sum[r], -sum[r]};
      f32x2 q0[P], e[P];
#pragma unroll
      for (int i = 0; i < P; ++i) q0[i] = t[r][i] * rs;
#pragma unroll
      for (int i = 0; i < P; ++i) e[i] = __builtin_elementwise_fma(q0[i], ns, t[r][i]);
#pragma unroll
      for (int i = 0; i < P; ++i) t[r][i] = __builtin_elementwise_fma(e[i], rs, q0[i]);
      qf_fake_quant2_n<P>(t[r], f2);
      if (__ballot(sum[r] != sum[r])) {          // rare: a NaN row
        if (sum[r] != sum[r]) {
#pragma unroll
          for (int i = 0; i < P; ++i) t[r][i] = f32x2{__builtin_nanf(""), __builtin_nanf("")};
        }
      }
      if (row0 + r >= rows) continue;
#pragma unroll
      for (int k = 0; k < NV; ++k)
        y[(row0 + r) * T4 + k * LPR + lane] = f32x4{t[r][2 * k].x, t[r][2 * k].y, t[r][2 * k + 1].x, t[r][2 * k + 1].y};
    }
  }
}

template <int LPR, int NV, int R>
__global__ __launch_bounds__(kBlock) void softmax_quant_k(const f32x4* __restrict__ s, f32x4* __restrict__ y, uint64_t rows,
                                                          const float* __restrict__ mask, uint64_t rows_per_mask,
                                                          float denom, tq_quantizer q1, tq_quantizer q2, int on1, int on2,
                                                          int allow_fast) {
  // R rows per lane group and step: R independent load -> reduce -> exp -> reduce -> store chains in flight
  if (allow_fast && softmax_fast_ok(q1, q2, on1, on2, denom)) {          // wave-uniform
    softmax_fast_body<LPR, NV, R>(s, y, rows, mask, rows_per_mask, denom, q1, q2);
    return;
  }
  constexpr int RPB = kBlock / LPR;
  constexpr uint32_t T4 = LPR * NV;              // float4 vectors per row
  const FusedQ f1 = make_fq(q1, on1), f2 = make_fq(q2, on2);
  const int lane = threadIdx.x % LPR, sub = threadIdx.x / LPR;
  for (uint64_t row0 = ((uint64_t)blockIdx.x * RPB + sub) * R; row0 < rows; row0 += (uint64_t)gridDim.x * RPB * R) {
    f32x4 in[R][NV], mk[R][NV];
    const uint64_t m0 = mask ? row0 / rows_per_mask : 0, rem0 = row0 - m0 * rows_per_mask;   // one 64-bit division per step
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint64_t row = row0 + r < rows ? row0 + r : rows - 1;        // clamp: tail rows recompute the last one
      const uint64_t mi = rem0 + r < rows_per_mask ? m0 : row / rows_per_mask;
      const f32x4* mrow = mask ? reinterpret_cast<const f32x4*>(mask + mi * (uint64_t)T4 * 4) : nullptr;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        in[r][k] = s[row * T4 + k * LPR + lane];
        mk[r][k] = mrow ? mrow[k * LPR + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    float v[R][NV][4], mx[R], sum[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      mx[r] = -__builtin_huge_valf();
#pragma unroll
      for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t = apply_q(in[r][k][j], f1) / denom;
          if (mask) t = t + mk[r][k][j];
          v[r][k][j] = t;
          mx[r] = fmaxf(mx[r], t);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) mx[r] = group_max<LPR>(mx[r]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      sum[r] = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[r][k][j] = expf(v[r][k][j] - mx[r]); sum[r] += v[r][k][j]; }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) sum[r] = group_sum<LPR>(sum[r]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (row0 + r >= rows) continue;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = apply_q(v[r][k][j] / sum[r], f2);
        y[(row0 + r) * T4 + k * LPR + lane] = o;
      }
    }
  }
}

}  // namespace tq

using namespace tq;

static int residual_tail(const char* who, const void* dense_out, const void* residual, void* y, int8_t* y_idx, uint64_t rows,
                         uint64_t d, int dtype, const tq_quantizer* q_dense, const tq_quantizer* q_sum, const float* weight,
                         const float* bias, float ln_eps, const tq_quantizer* q_out, int affine_only, tq_stream_t stream) {
  if (rows == 0) return TQ_OK;
  TQ_REQUIRE(dense_out && residual && y && weight && bias, "%s: NULL pointer", who);
  TQ_REQUIRE(dtype == TQ_F32 || dtype == TQ_BF16 || dtype == TQ_F16, "%s: bad dtype %d", who, dtype);
  TQ_REQUIRE(aligned16(dense_out) && aligned16(residual) && aligned16(y), "%s: 16-byte alignment required", who);
  TQ_REQUIRE(y_idx == nullptr || (q_out != nullptr && !q_out->symmetric && q_out->n_bits <= 8 && (reinterpret_cast<uintptr_t>(y_idx) & 7u) == 0),
             "%s: y_idx needs an asymmetric <= 8-bit output quantizer and 8-byte alignment", who);
  for (const tq_quantizer* q : {q_dense, q_sum, q_out})
    if (q != nullptr) {
      if (int e = check_quantizer(q, rows * d, who)) return e;
      TQ_REQUIRE(q->n_params == 1, "%s: per-tensor quantizers only", who);
    }
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case TQ_F32: return launch_res_ln<TQ_F32>(dense_out, residual, y, y_idx, rows, d, weight, bias, ln_eps, q_dense, q_sum, q_out, affine_only, st);
    case TQ_BF16: return launch_res_ln<TQ_BF16>(dense_out, residual, y, y_idx, rows, d, weight, bias, ln_eps, q_dense, q_sum, q_out, affine_only, st);
    default: return launch_res_ln<TQ_F16>(dense_out, residual, y, y_idx, rows, d, weight, bias, ln_eps, q_dense, q_sum, q_out, affine_only, st);
  }
}

extern "C" int tq_residual_layernorm_quant_fwd(const void* dense_out, const void* residual, void* y, int8_t* y_idx, uint64_t rows,
                                               uint64_t d, int dtype, const tq_quantizer* q_dense,
                                               const tq_quantizer* q_sum, const float* ln_weight, const float* ln_bias,
                                               float ln_eps, const tq_quantizer* q_out, tq_stream_t stream) {
  return residual_tail("tq_residual_layernorm_quant_fwd", dense_out, residual, y, y_idx, rows, d, dtype, q_dense, q_sum, ln_weight,
                       ln_bias, ln_eps, q_out, 0, stream);
}

extern "C" int tq_embeddings_layernorm_quant_fwd(const float* word_table, uint64_t word_rows, const int64_t* word_ids,
                                                 const float* type_table, uint64_t type_rows, const int64_t* type_ids,
                                                 const float* pos_table, uint64_t pos_rows, const int64_t* pos_ids, float* y,
                                                 int8_t* y_idx, uint64_t rows, uint64_t d, const tq_quantizer* q_sum1,
                                                 const tq_quantizer* q_sum2, const float* ln_weight, const float* ln_bias,
                                                 float ln_eps, const tq_quantizer* q_out, int32_t* bad_ids,
                                                 tq_stream_t stream) {
  const char* who = "tq_embeddings_layernorm_quant_fwd";
  if (rows == 0) return TQ_OK;
  TQ_REQUIRE(word_table && type_table && pos_table && word_ids && type_ids && pos_ids && y && ln_weight && ln_bias, "%s: NULL pointer", who);
  TQ_REQUIRE(word_rows >= 1 && type_rows >= 1 && pos_rows >= 1, "%s: empty table", who);
  TQ_REQUIRE(aligned16(word_table) && aligned16(type_table) && aligned16(pos_table) && aligned16(y), "%s: 16-byte alignment required", who);
  TQ_REQUIRE(d % 4 == 0, "%s: row length %llu is not a whole number of 16-byte vectors", who, (unsigned long long)d);
  TQ_REQUIRE(y_idx == nullptr || (q_out != nullptr && !q_out->symmetric && q_out->n_bits <= 8 && (reinterpret_cast<uintptr_t>(y_idx) & 7u) == 0),
             "%s: y_idx needs an asymmetric <= 8-bit output quantizer and 8-byte alignment", who);
  for (const tq_quantizer* q : {q_sum1, q_sum2, q_out})
    if (q != nullptr) {
      if (int e = check_quantizer(q, rows * d, who)) return e;
      TQ_REQUIRE(q->n_params == 1, "%s: per-tensor quantizers only", who);
    }
  TQ_REQUIRE(bad_ids == nullptr || (reinterpret_cast<uintptr_t>(bad_ids) & 3u) == 0, "%s: bad_ids must be 4-byte aligned", who);
  const EmbArgs emb{word_ids, type_ids, pos_ids, reinterpret_cast<const u32x4*>(type_table), word_rows, type_rows, pos_rows, bad_ids};
  return launch_res_ln<TQ_F32>(word_table, pos_table, y, y_idx, rows, d, ln_weight, ln_bias, ln_eps, q_sum1, q_sum2, q_out, 0,
                               static_cast<hipStream_t>(stream), &emb);
}

extern "C" int tq_residual_nonorm_quant_fwd(const void* dense_out, const void* residual, void* y, int8_t* y_idx, uint64_t rows,
                                            uint64_t d, int dtype, const tq_quantizer* q_dense, const tq_quantizer* q_sum,
                                            const float* weight, const float* bias, const tq_quantizer* q_out,
                                            tq_stream_t stream) {
  return residual_tail("tq_residual_nonorm_quant_fwd", dense_out, residual, y, y_idx, rows, d, dtype, q_dense, q_sum, weight, bias,
                       0.0f, q_out, 1, stream);
}

extern "C" int tq_scores_softmax_quant_fwd(const float* scores, float* probs, uint64_t rows, uint64_t cols,
                                           const float* mask, uint64_t rows_per_mask, float denom,
                                           const tq_quantizer* q_scores, const tq_quantizer* q_probs, tq_stream_t stream) {
  if (rows == 0) return TQ_OK;
  TQ_REQUIRE(scores && probs, "tq_scores_softmax_quant_fwd: NULL pointer");
  TQ_REQUIRE(aligned16(scores) && aligned16(probs) && (mask == nullptr || aligned16(mask)),
             "tq_scores_softmax_quant_fwd: 16-byte alignment required");
  TQ_REQUIRE(mask == nullptr || (rows_per_mask >= 1 && rows % rows_per_mask == 0), "tq_scores_softmax_quant_fwd: bad mask layout");
  if (mask == nullptr) rows_per_mask = rows;      // unused without a mask; keeps the kernels' row -> mask arithmetic away from / 0
  TQ_REQUIRE(denom != 0.0f, "tq_scores_softmax_quant_fwd: denom == 0");
  for (const tq_quantizer* q : {q_scores, q_probs})
    if (q != nullptr) {
      if (int e = check_quantizer(q, rows * cols, "tq_scores_softmax_quant_fwd")) return e;
      TQ_REQUIRE(q->n_params == 1, "tq_scores_softmax_quant_fwd: per-tensor quantizers only");
    }
  const tq_quantizer none{};
  const tq_quantizer &c1 = q_scores ? *q_scores : none, &c2 = q_probs ? *q_probs : none;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const auto sv = reinterpret_cast<const f32x4*>(scores);
  auto yv = reinterpret_cast<f32x4*>(probs);
#define TQ_SM(LPR, NV)                                                                                          \
  if (cols == (uint64_t)(LPR) * (NV) * 4) {                                                                     \
    static const int r_env = tuning("TQ_SM_R", 0);                                                              \
    const int allow_fast = tuning("TQ_SM_FAST", 1);      /* read per call: the parity test flips it */            \
    /* rows in flight per lane group: as many as the registers allow, while the grid still covers the chip twice */ \
    int want = r_env ? r_env : 4;                                                                               \
    while (!r_env && want > 1 && rows / (kBlock / (LPR) * want) < 512) want >>= 1;                              \
    const int R = (want == 4 && (NV) == 1) ? 4 : ((want >= 2 && (NV) <= 2) ? 2 : 1);                            \
    const unsigned rpb = kBlock / (LPR) * R;                                                                    \
    const unsigned grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>(ceil_div(rows, rpb), 1), 1u << 20);   \
    if (R == 4 && (NV) == 1)                                                                                    \
      hipLaunchKernelGGL((softmax_quant_k<LPR, NV, 4>), dim3(grid), dim3(kBlock), 0, st, sv, yv, rows, mask, rows_per_mask, \
                         denom, c1, c2, q_scores != nullptr, q_probs != nullptr, allow_fast);                               \
    else if (R == 2 && (NV) <= 2)                                                                               \
      hipLaunchKernelGGL((softmax_quant_k<LPR, NV, 2>), dim3(grid), dim3(kBlock), 0, st, sv, yv, rows, mask, rows_per_mask, \
                         denom, c1, c2, q_scores != nullptr, q_probs != nullptr, allow_fast);                               \
    else                                                                                                        \
      hipLaunchKernelGGL((softmax_quant_k<LPR, NV, 1>), dim3(grid), dim3(kBlock), 0, st, sv, yv, rows, mask, rows_per_mask, \
                         denom, c1, c2, q_scores != nullptr, q_probs != nullptr, allow_fast);                               \
    return check_launch("softmax_quant_k");                                                                     \
  }
  TQ_SM(8, 1) TQ_SM(16, 1) TQ_SM(32, 1) TQ_SM(64, 1) TQ_SM(64, 2) TQ_SM(64, 4)
#undef TQ_SM
  return set_error(TQ_EUNSUPPORTED, "tq_scores_softmax_quant_fwd: row length %llu unsupported (32..1024, power of two)",
                   (unsigned long long)cols);
}
