// (f3) Fused integer Linear + bias + activation + fake-quant for gfx950 MFMA.
//
// In the reference a quantized Linear is `F.linear(Q(x), Q(W), b)` on DEQUANTISED fp32 tensors
// followed by an optional activation function and the output quantizer (quantization/hijacker.py:
// 66-116, autoquant_utils.py:16-21).  With fixed ranges both operands live on integer grids,
//     x = s_x (a - z_x),  a in [0, 2^n)        W = s_w w,  w in [-2^(n-1), 2^(n-1))
// so the GEMM is an exact integer contraction:
//     out[m, n] = s_x s_w[n] ( sum_k a'[m,k] w[n,k] + (128 - z_x) rowsum_w[n] ) + b[n],   a' = a - 128
// which runs on the i8 matrix cores (v_mfma_i32_16x16x64_i8, ~2x the bf16 rate, ~32x the fp32 rate)
// with i32 accumulation (|acc| + |correction| <= 128*(128+255)*K < 2^31 for K <= 16384).  The result differs from the
// reference's fp32 simulation only by the simulation's own fp32 accumulation round-off (it is the
// exact value the simulation approximates); bias, activation and the output quantizer run in the
// epilogue on the accumulator registers: the [M, N] pre-activation tensor never touches HBM.
//
// MFMA fragment layout used throughout: lane l supplies row l & 15 and the 16 k-bytes of group l >> 4
// (any k permutation is legal as long as both operands use the same one).  The W tile is the FIRST
// MFMA operand so that a lane's 4 accumulator registers are 4 consecutive output features of one
// token: 16-byte stores.  Two kernels: LDS-staged (fast path, see below) and an LDS-free fallback for
// shapes that are not multiples of 64 / 128.
#include <algorithm>
#include <type_traits>

#include "tq_device.h"
#include "tq_host.h"

namespace tq {

typedef int v4i __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_TANH = 3 };

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {   // wave-uniform
    case ACT_RELU: return v > 0.0f ? v : 0.0f;
    case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));   // nn.GELU() (erf form)
    case ACT_TANH: return tanhf(v);
    default: return v;
  }
}

struct LinArgs {
  const int8_t* x;        // [M, K] activation indices - 128
  const int8_t* w;        // [N, K] weight indices
  const int32_t* w_rowsum;
  const float* bias;      // [N] or null
  void* y;                // [M, N]
  int8_t* y_idx;          // optional [M, N] int8(index - 128) of y (needs q_out), or null
  uint32_t M, N, K;
  const float* x_delta;   // per-tensor input quantizer
  const float* x_zero_float;
  float x_eps;
  int x_n_bits;
  const float* w_delta;   // [1] or [N]
  uint32_t w_n_params;
  float w_eps;
  int act;
  int has_q;
  tq_quantizer q_out;     // group 0
  tq_quantizer q_out1, q_out2;   // groups 1, 2 of a grouped launch (Q | K | V stacked along N)
  uint32_t group_cols;    // output columns per group (N for a plain launch); multiple of 64
  int fast_epi;           // 0 forces the generic epilogue (TQ_I8_FAST_EPI=0: A/B and tests)
  int dbg;                // breakdown builds only (-DTQ_I8_DBG_BUILD, env TQ_I8_DBG): 1 no epilogue, 2 no operand loads, 4 no MFMA
  // optional NoNorm tail fused behind the output quantizer (MobileBERT; models/quantized_mobilebert.py:58-72, 287-352):
  //   tail 1:  y = Q_t2( Q_out(v) * nn_w + nn_b )                       bottleneck Linear -> NoNorm
  //   tail 2:  y = Q_t2( Q_t1( Q_out(v) + residual ) * nn_w + nn_b )    Linear -> + residual -> NoNorm
  // y / y_idx are then the outputs of Q_t2.  mul and add separate, like the reference's `x * weight + bias`.
  int tail;
  const float* residual;  // [M, N] fp32 (tail 2)
  const float* nn_w;      // [N] fake-quantized NoNorm weight
  const float* nn_b;      // [N] fake-quantized NoNorm bias
  tq_quantizer q_t1, q_t2;
  int on_t1, on_t2;
  // grouped launch WITH tail 1 (MobileBERT's two input bottlenecks -- and the value Linear -- read the same tensor): the
  // NoNorm output quantizers of groups 1 and 2, and the outputs of group g as a tensor of their own, [M, group_cols] at
  // y + g * M * group_cols
  tq_quantizer q_t2b, q_t2c;
  int split_out;
  // optional staircase table of act + q_out (tq_act_stair_build; LDS kernels only): replaces the activation and the
  // quantizer's quotient in the epilogue when its header says it is exact
  const float* stair;
  uint32_t stair_bins;
};

struct StairRef {                // the table as the epilogue sees it: entries in LDS, geometry in scalars
  const u32x2* tab;
  float inv_w, c0, nbm1;
};

// Element offset of output (row, n) in y / y_idx; g = out_group(p, n0) of the wave's first column, computed ONCE per wave
// (a tile never straddles two groups) -- with the division inside, the epilogue loops were no longer unrolled and the
// accumulators of the 128 x 128 tile kernel moved to scratch memory.
__device__ __forceinline__ uint32_t out_group(const LinArgs& p, uint32_t n0) { return p.split_out ? n0 / p.group_cols : 0; }
__device__ __forceinline__ size_t out_at(const LinArgs& p, uint32_t g, uint32_t row, uint32_t n) {
  if (!p.split_out) return (size_t)row * p.N + n;
  return (size_t)g * p.M * p.group_cols + (size_t)row * p.group_cols + (n - g * p.group_cols);
}

// ---- epilogue: zero-point correction, scales, bias, activation, output quantizer ----------------------
// The epilogue is the expensive half of this kernel at BERT's K = 768: every output element costs one GEMM column
// of 768 MACs = 0.75 MFMA-lane-cycles, but (with the IEEE division, libm's erff and a per-element activation switch)
// ~100 VALU instructions and 11 branches.  The fast form below is branch-free and packed (2 elements per instruction
// where the ISA allows): ~27 issue slots per element with GELU + quantizer.

// GELU for NP value pairs.  Round 4: ONE fit for the whole axis instead of the ROCm device library's two (|t| < 1: odd
// polynomial, else 1 - exp(-(t + t p(t)))), which had to be evaluated both and selected: with t = |v| / sqrt 2,
//     erfc(t) = 2^(-t Q(t)),  Q of degree 7 fitted on [0, 4] for a uniform ABSOLUTE error of erf (1.6e-8 fit + fp32 evaluation:
//     8.5e-8 in total; beyond t = 4 the exponent keeps growing -- leading coefficient > 0 -- and erfc underflows to the 0 it
//     is in fp32 from t = 3.92 on),  erf(a) = copysign(1 - erfc(t), a),  GELU = (0.5 v) (1 + erf).
// Against nn.GELU() evaluated in float64 the result is as close as the reference's own fp32 evaluation (max 4.7e-7 vs
// 4.5e-7 absolute on N(0, 1.5^2) pre-activations; tools/tuning/gelu_fit.py), at 13 packed + 6 scalar issue slots per pair instead of 24 + 10.
// exp2 is the hardware v_exp_f32 (<= 1 ulp); oracle/tq_int_oracle.c holds the identical formula with exp2f, and the
// integer-path tests hold the two to <= 1 grid step on <= 1e-5 of the outputs behind an 8-bit quantizer.
template <int NP>
__device__ __forceinline__ void gelu_erf_n(f32x2 (&v)[NP]) {
  f32x2 a[NP], t[NP], q[NP];
  const f32x2 one = {1.0f, 1.0f};
  auto k2 = [](float c) { return f32x2{c, c}; };
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    a[i] = v[i] * k2(0.70710678118654752440f);
    t[i] = __builtin_elementwise_abs(a[i]);
  }
  // q = -Q(t): the coefficients carry the sign, so that 2^(q t) is erfc directly
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = __builtin_elementwise_fma(t[i], k2(-4.5358574425335974e-05f), k2(0.00044550743768922985f));
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = __builtin_elementwise_fma(t[i], q[i], k2(-0.0014894409105181694f));
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = __builtin_elementwise_fma(t[i], q[i], k2(-0.0007746309274807572f));
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = __builtin_elementwise_fma(t[i], q[i], k2(0.02825368195772171f));
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = __builtin_elementwise_fma(t[i], q[i], k2(-0.14848162233829498f));
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = __builtin_elementwise_fma(t[i], q[i], k2(-0.9184163808822632f));
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = __builtin_elementwise_fma(t[i], q[i], k2(-1.6279085874557495f));
#pragma unroll
  for (int i = 0; i < NP; ++i) q[i] = q[i] * t[i];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    f32x2 e = {__builtin_amdgcn_exp2f(q[i].x), __builtin_amdgcn_exp2f(q[i].y)};     // erfc(t)
    e = one - e;                                                                      // erf(t) >= 0
    f32x2 r;
    r.x = __builtin_copysignf(e.x, a[i].x);
    r.y = __builtin_copysignf(e.y, a[i].y);
    v[i] = (v[i] * k2(0.5f)) * (one + r);                              // nn.GELU(): x * 0.5 * (1 + erf(x / sqrt(2)))
  }
}

template <int YDT>
__device__ __forceinline__ void store_y4(void* y, size_t at, f32x2 lo, f32x2 hi) {
  if (YDT == TQ_F32) {
    *reinterpret_cast<f32x4*>(static_cast<float*>(y) + at) = f32x4{lo.x, lo.y, hi.x, hi.y};
  } else {
    u32x2 pk;
    pk[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));
    pk[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2));
    *reinterpret_cast<u32x2*>(static_cast<uint16_t*>(y) + at) = pk;
  }
}

// Staged output of one pass (PR token rows x WTN output features of a wave, parked in its private LDS area): row-contiguous
// stores, every instruction covering whole 128-byte lines; non-temporal for y (not read again by this kernel).
// Wave-private staging: program order (+ the compiler's lgkmcnt) is all the synchronisation needed.
template <int WTN, int PR, int YDT>
__device__ __forceinline__ void stage_flush(const LinArgs& p, const int8_t* ystage, const int8_t* istage, uint32_t n0, uint32_t mrow0,
                                            int lane, bool want_idx) {
  constexpr int ES = YDT == TQ_F32 ? 4 : 2;
  constexpr int YP = WTN * ES + 16, IP = WTN + 16;
  const uint32_t og = out_group(p, n0);
  if (p.y != nullptr) {
    constexpr int LPR = WTN * ES / 16, RPI = 64 / LPR;   // lanes per row, rows per store instruction
#pragma unroll
    for (int t = 0; t < (PR + RPI - 1) / RPI; ++t) {
      const int row = t * RPI + lane / LPR;
      if (RPI <= PR || row < PR) {
        const u32x4 d = *reinterpret_cast<const u32x4*>(ystage + row * YP + (lane % LPR) * 16);
        __builtin_nontemporal_store(d, reinterpret_cast<u32x4*>(static_cast<int8_t*>(p.y) +
                                                                out_at(p, og, mrow0 + row, n0) * ES + (lane % LPR) * 16));
      }
    }
  }
  if (want_idx) {
    constexpr int LPR = WTN / 16, RPI = 64 / LPR;
#pragma unroll
    for (int t = 0; t < (PR + RPI - 1) / RPI; ++t) {
      const int row = t * RPI + lane / LPR;
      if (RPI <= PR || row < PR) {
        const u32x4 d = *reinterpret_cast<const u32x4*>(istage + row * IP + (lane % LPR) * 16);
        *reinterpret_cast<u32x4*>(p.y_idx + out_at(p, og, mrow0 + row, n0) + (lane % LPR) * 16) = d;
      }
    }
  }
}

// Fast form: ACT in {none, relu, gelu}; HASQ needs a quantizer the exact-quotient path covers (QF::ok).
// `stage` != nullptr (LDS kernel): the wave parks its results in a private LDS region, 32 token rows per pass, and
// writes them out row-contiguously -- every store instruction covers whole 128-byte lines (fp32 y: 4 rows x 256 B)
// instead of 16 rows x 64 B (y) or 16 rows x 16 B (indices) straight from the MFMA accumulator layout -- with
// non-temporal stores (y is not read again by this kernel; W and X keep the L2).  Measured at M = 8192, N = 3072,
// K = 768, fp32 y: 48.4 -> 34.5 us for the GEMM + plain store.
// FLUSH = false (chained feed-forward blocks): the int8 indices of the single pass stay in the wave's staging area (at
// stage + 16 * YP, rows of pitch IP; the y part is not written -- y = scale * (index - zp) exactly, rebuilt by the caller).
template <int NI, int MI, int YDT, int ACT, bool HASQ, bool STAGED, int TAIL = 0, bool FLUSH = true>
__device__ __forceinline__ void linear_epilogue_fast(const LinArgs& p, v4i (&acc)[NI][MI], uint32_t n0, uint32_t m0, int r16,
                                                     int kg, const QF& qf, int shift, float sx, int8_t* stage,
                                                     const float* cst, const QF& qf1 = QF{}, const QF& qf2 = QF{},
                                                     int cs = 2 * NI * 16 /* floats between the arrays of `cst` */,
                                                     const f32x4 (*res_pre)[MI] = nullptr /* residual fetched early */) {
  constexpr int JP = STAGED ? (MI >= 2 ? 2 : 1) : MI;   // j tiles per pass (staged: 32 token rows, 16 for a 16-row tile)
  constexpr int PR = JP * 16;                           // token rows per staged pass
  constexpr int NP = 2 * JP;
  constexpr int WTN = NI * 16;                          // output features of this wave
  constexpr int ES = YDT == TQ_F32 ? 4 : 2;
  constexpr int YP = WTN * ES + 16, IP = WTN + 16;      // staging row pitches (y, indices): + 16 B against bank conflicts
  // the quantizer whose grid y (and y_idx) live on: Q_out, or the tail's last quantizer
  const bool fin_q = TAIL ? (p.on_t2 != 0) : HASQ;
  const float zfin = TAIL ? qf2.zp : qf.zp;
  const f32x2 zpb = {zfin, zfin};
  const int lane = kg * 16 + r16;
  int8_t* ystage = stage;
  int8_t* istage = stage + PR * YP;
  static_assert(FLUSH || (STAGED && MI == JP), "results kept in the staging area: one pass");
  const uint32_t og = out_group(p, n0);
  const bool want_idx = fin_q && (p.y_idx != nullptr || !FLUSH);
  const bool want_y = FLUSH && p.y != nullptr;       // (FLUSH = false: y is rebuilt from the indices by the caller)
#pragma unroll
  for (int h = 0; h < MI / JP; ++h) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const uint32_t n = n0 + i * 16 + kg * 4;            // this lane's 4 consecutive output features
      f32x2 sw[2], bs[2];
      int rs[4];
      if (STAGED) {                                        // per-column constants prepared in LDS by the kernel prologue
        const int col = i * 16 + kg * 4;
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(cst + col);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(cst + cs + col);
        const v4i r4 = *reinterpret_cast<const v4i*>(cst + 2 * cs + col);
        sw[0] = f32x2{s4.x, s4.y}; sw[1] = f32x2{s4.z, s4.w};
        bs[0] = f32x2{b4.x, b4.y}; bs[1] = f32x2{b4.z, b4.w};
        rs[0] = r4.x; rs[1] = r4.y; rs[2] = r4.z; rs[3] = r4.w;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dw = p.w_delta[p.w_n_params == 1 ? 0 : n + r];
          sw[r >> 1][r & 1] = sx * (dw < p.w_eps ? p.w_eps : dw);
          bs[r >> 1][r & 1] = p.bias ? p.bias[n + r] : 0.0f;
          rs[r] = p.w_rowsum[n + r] * shift;
        }
      }
      f32x2 v[NP];
#pragma unroll
      for (int jj = 0; jj < JP; ++jj) {
        const int j = h * JP + jj;
        const f32x2 lo = {(float)(acc[i][j][0] + rs[0]), (float)(acc[i][j][1] + rs[1])};
        const f32x2 hi = {(float)(acc[i][j][2] + rs[2]), (float)(acc[i][j][3] + rs[3])};
        v[2 * jj] = lo * sw[0] + bs[0];                    // separate mul and add as in the reference (no contraction)
        v[2 * jj + 1] = hi * sw[1] + bs[1];
      }
      if (ACT == ACT_GELU) gelu_erf_n<NP>(v);
      if (ACT == ACT_RELU) {
#pragma unroll
        for (int e = 0; e < NP; ++e) v[e] = f32x2{v[e].x > 0.0f ? v[e].x : 0.0f, v[e].y > 0.0f ? v[e].y : 0.0f};
      }
      f32x2 hq[NP];
      if (HASQ) {
        qf_round2_n<NP>(v, qf, hq);
#pragma unroll
        for (int e = 0; e < NP; ++e) v[e] = qf_dequant2(hq[e], qf);
      }
      if (TAIL == 2) {                                      // + residual, then the sum quantizer
#pragma unroll
        for (int jj = 0; jj < JP; ++jj) {
          const f32x4 r4 = res_pre != nullptr ? res_pre[i][h * JP + jj]
                                              : *reinterpret_cast<const f32x4*>(p.residual + (size_t)(m0 + (h * JP + jj) * 16 + r16) * p.N + n);
          v[2 * jj] = v[2 * jj] + f32x2{r4.x, r4.y};
          v[2 * jj + 1] = v[2 * jj + 1] + f32x2{r4.z, r4.w};
        }
        if (p.on_t1) {
          qf_round2_n<NP>(v, qf1, hq);
#pragma unroll
          for (int e = 0; e < NP; ++e) v[e] = qf_dequant2(hq[e], qf1);
        }
      }
      if (TAIL >= 1) {                                      // NoNorm affine (mul, then add), then its output quantizer
        f32x2 nw[2], nb[2];
        if (STAGED) {
          const int col = i * 16 + kg * 4;
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(cst + 3 * cs + col);
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(cst + 4 * cs + col);
          nw[0] = f32x2{w4.x, w4.y}; nw[1] = f32x2{w4.z, w4.w};
          nb[0] = f32x2{b4.x, b4.y}; nb[1] = f32x2{b4.z, b4.w};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) { nw[r >> 1][r & 1] = p.nn_w[n + r]; nb[r >> 1][r & 1] = p.nn_b[n + r]; }
        }
#pragma unroll
        for (int jj = 0; jj < JP; ++jj) {
          v[2 * jj] = v[2 * jj] * nw[0] + nb[0];
          v[2 * jj + 1] = v[2 * jj + 1] * nw[1] + nb[1];
        }
        if (p.on_t2) {
          qf_round2_n<NP>(v, qf2, hq);
#pragma unroll
          for (int e = 0; e < NP; ++e) v[e] = qf_dequant2(hq[e], qf2);
        }
      }
#pragma unroll
      for (int jj = 0; jj < JP; ++jj) {
        const int j = h * JP + jj;
        uint32_t w = 0;
        if (want_idx) {                                   // int8(index - 128): u8 index with the top bit flipped
          const f32x2 a = hq[2 * jj] + zpb, b = hq[2 * jj + 1] + zpb;
          w = __builtin_amdgcn_cvt_pk_u8_f32(a.x, 0, w);
          w = __builtin_amdgcn_cvt_pk_u8_f32(a.y, 1, w);
          w = __builtin_amdgcn_cvt_pk_u8_f32(b.x, 2, w);
          w = __builtin_amdgcn_cvt_pk_u8_f32(b.y, 3, w) ^ 0x80808080u;
        }
        if (STAGED) {
          const int row = jj * 16 + r16, col = i * 16 + kg * 4;
          if (want_idx) *reinterpret_cast<uint32_t*>(istage + row * IP + col) = w;
          if (want_y) store_y4<YDT>(ystage + row * YP, col, v[2 * jj], v[2 * jj + 1]);
        } else {
          const size_t at = out_at(p, og, m0 + j * 16 + r16, n);
          if (want_idx) *reinterpret_cast<uint32_t*>(p.y_idx + at) = w;
          if (p.y != nullptr) store_y4<YDT>(p.y, at, v[2 * jj], v[2 * jj + 1]);
        }
      }
    }
    if (STAGED && FLUSH) stage_flush<WTN, PR, YDT>(p, ystage, istage, n0, m0 + h * PR, lane, want_idx);
  }
}

// Staircase form (LDS kernels, no tail): h = Q(act(v)) - zp straight from the table of csrc/tq_stair.hip -- per output
// one fma + clamp + convert for the bin, one 8-byte LDS read, compare, select -- in two phases per pass so that the table
// reads of ALL the pass's outputs (32 per lane for a 64 x 64 wave tile) are in flight together: with one or two waves per
// SIMD nothing else hides an LDS round trip, and issued group by group the reads cost more than the arithmetic they replace.
template <int NI, int MI, int YDT>
__device__ __forceinline__ void linear_epilogue_stair(const LinArgs& p, v4i (&acc)[NI][MI], uint32_t n0, uint32_t m0, int r16,
                                                      int kg, const QF& qf, int8_t* stage, const float* cst, int cs,
                                                      const StairRef& sr) {
  constexpr int JP = MI >= 2 ? 2 : 1, PR = JP * 16, NP = 2 * JP, WTN = NI * 16;
  constexpr int ES = YDT == TQ_F32 ? 4 : 2;
  constexpr int YP = WTN * ES + 16, IP = WTN + 16;
  const f32x2 zpb = {qf.zp, qf.zp};
  const int lane = kg * 16 + r16;
  int8_t* ystage = stage;
  int8_t* istage = stage + PR * YP;
  const bool want_idx = p.y_idx != nullptr;
#pragma unroll
  for (int h = 0; h < MI / JP; ++h) {
    f32x2 v[NI][NP];
    u32x2 e0[NI][NP], e1[NI][NP];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = i * 16 + kg * 4;
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(cst + col);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(cst + cs + col);
      const v4i r4 = *reinterpret_cast<const v4i*>(cst + 2 * cs + col);
      const f32x2 sw0 = {s4.x, s4.y}, sw1 = {s4.z, s4.w}, bs0 = {b4.x, b4.y}, bs1 = {b4.z, b4.w};
#pragma unroll
      for (int jj = 0; jj < JP; ++jj) {
        const int j = h * JP + jj;
        const f32x2 lo = {(float)(acc[i][j][0] + r4.x), (float)(acc[i][j][1] + r4.y)};
        const f32x2 hi = {(float)(acc[i][j][2] + r4.z), (float)(acc[i][j][3] + r4.w)};
        v[i][2 * jj] = lo * sw0 + bs0;                     // separate mul and add as in the reference (no contraction)
        v[i][2 * jj + 1] = hi * sw1 + bs1;
      }
#pragma unroll
      for (int e = 0; e < NP; ++e) {
        e0[i][e] = sr.tab[stair_bin(v[i][e].x, sr.inv_w, sr.c0, sr.nbm1)];
        e1[i][e] = sr.tab[stair_bin(v[i][e].y, sr.inv_w, sr.c0, sr.nbm1)];
      }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      f32x2 hq[NP], y[NP];
#pragma unroll
      for (int e = 0; e < NP; ++e) {
        const uint32_t t0 = e0[i][e].x, p0 = e0[i][e].y, t1 = e1[i][e].x, p1 = e1[i][e].y;
        hq[e] = f32x2{stair_pick(v[i][e].x, bits_to_f32(t0), p0), stair_pick(v[i][e].y, bits_to_f32(t1), p1)};
        y[e] = qf_dequant2(hq[e], qf);
      }
#pragma unroll
      for (int jj = 0; jj < JP; ++jj) {
        const int row = jj * 16 + r16, col = i * 16 + kg * 4;
        if (want_idx) {                                   // int8(index - 128): u8 index with the top bit flipped
          const f32x2 a = hq[2 * jj] + zpb, b = hq[2 * jj + 1] + zpb;
          uint32_t w = 0;
          w = __builtin_amdgcn_cvt_pk_u8_f32(a.x, 0, w);
          w = __builtin_amdgcn_cvt_pk_u8_f32(a.y, 1, w);
          w = __builtin_amdgcn_cvt_pk_u8_f32(b.x, 2, w);
          w = __builtin_amdgcn_cvt_pk_u8_f32(b.y, 3, w) ^ 0x80808080u;
          *reinterpret_cast<uint32_t*>(istage + row * IP + col) = w;
        }
        if (p.y != nullptr) store_y4<YDT>(ystage + row * YP, col, y[2 * jj], y[2 * jj + 1]);
      }
    }
    stage_flush<WTN, PR, YDT>(p, ystage, istage, n0, m0 + h * PR, lane, want_idx);
  }
}

// Generic form (tanh, quantizers outside the exact-quotient path): IEEE division, libm activation.
template <int NI, int MI, int YDT>
__device__ __forceinline__ void linear_epilogue_generic(const LinArgs& p, v4i (&acc)[NI][MI], uint32_t n0, uint32_t m0, int r16,
                                                     int kg, const QP& qo, int shift, float sx,
                                                     const f32x4 (*res_pre)[MI] = nullptr /* residual in registers */,
                                                     int8_t* keep = nullptr /* staging area: the indices stay there (MI == 1) */) {
  constexpr bool BIG = NI * MI >= 16;          // the 128 x 128 tile kernel: tq_device.h effective_scale's select form (see there)
  const uint32_t og = out_group(p, n0);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const uint32_t n = n0 + i * 16 + kg * 4;
    float sw[4], bs[4];
    int rs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dw = p.w_delta[p.w_n_params == 1 ? 0 : n + r];
      sw[r] = sx * (dw < p.w_eps ? p.w_eps : dw);
      bs[r] = p.bias ? p.bias[n + r] : 0.0f;
      rs[r] = p.w_rowsum[n + r] * shift;
    }
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const size_t at = out_at(p, og, m0 + j * 16 + r16, n), rat = (size_t)(m0 + j * 16 + r16) * p.N + n;
      float o[4];
      struct alignas(4) { int8_t e[4]; } oi4 = {{0, 0, 0, 0}};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = (float)(acc[i][j][r] + rs[r]) * sw[r] + bs[r];
        v = apply_act(v, p.act);
        if (p.has_q) {
          const float xi = q_index(v, qo);
          oi4.e[r] = (int8_t)((int)xi - 128);
          v = q_dequant(xi, qo);
        }
        if (p.tail == 2) {
          v = v + (res_pre != nullptr ? res_pre[i][j][r] : p.residual[rat + r]);
          if (p.on_t1) v = q_dequant(q_index(v, make_qp<BIG>(p.q_t1, 0)), make_qp<BIG>(p.q_t1, 0));
        }
        if (p.tail >= 1) {
          v = v * p.nn_w[n + r] + p.nn_b[n + r];
          if (p.on_t2) {
            const QP q2 = make_qp<BIG>(og == 1 ? p.q_t2b : (og == 2 ? p.q_t2c : p.q_t2), 0);
            const float xi = q_index(v, q2);
            oi4.e[r] = (int8_t)((int)xi - 128);
            v = q_dequant(xi, q2);
          }
        }
        o[r] = v;
      }
      if (keep != nullptr) {                      // indices only, in the layout of the staged fast epilogue (16 rows, pitch IP)
        constexpr int ES = YDT == TQ_F32 ? 4 : 2, YP = NI * 16 * ES + 16, IP = NI * 16 + 16;
        const int row = j * 16 + r16, col = i * 16 + kg * 4;
        *reinterpret_cast<uint32_t*>(keep + 16 * YP + row * IP + col) = __builtin_bit_cast(uint32_t, oi4);
      } else {
        if (p.y_idx != nullptr) *reinterpret_cast<uint32_t*>(p.y_idx + at) = __builtin_bit_cast(uint32_t, oi4);
        if (p.y != nullptr) store_y4<YDT>(p.y, at, f32x2{o[0], o[1]}, f32x2{o[2], o[3]});
      }
    }
  }
}

// Everything the epilogue reads through pointers -- the quantizers' range buffers, the input scale / zero point -- as
// values.  The LDS kernels build it BEFORE the main loop: the scalar loads (~1 us cold each, partly dependent) then
// overlap with the operand fetches instead of sitting exposed between the last MFMA and the first store (measured on
// the feed-forward block kernel: 2.8 us of its 7 us were this epilogue's load latency).
struct EpiCtx {
  QP qo;
  QF qf, qf1, qf2;
  float sx;
  int shift;
  bool fast;
  bool stair;                    // the staircase table is present and exact (its header's verdict, read on the device)
  float st_inv_w, st_c0, st_nbm1;
};

// In two halves (tq_device.h load_qraw): epilogue_fetch issues every read the epilogue parameters need -- input quantizer,
// output quantizer of this block's column group, the tail's quantizers, the staircase header -- as independent loads;
// epilogue_finish turns the values into the context.  The LDS kernel calls the first half right after issuing its first
// operand slabs, so that parameters and operands share one memory round trip (round 5: the dependent chain
// x_delta -> q_out.delta -> q_out.zero_float -> ... in front of the first slab was ~15 round trips, about half of the
// 6-8 us these launches take at BERT's inference shapes).
struct EpiRaw {
  float dx, zx;
  QRaw qo, q1, q2;
  float st[4];
  tq_quantizer qsel;             // the output quantizer of this column group (by value: stays in scalar registers)
  tq_quantizer t2sel;            // the tail's last quantizer of this column group
};

// GROUPED = false: the caller knows the launch has ONE column group (feed-forward kernels): no run-time selection
// between the argument block's quantizer slots.
template <bool WITH_TAIL, bool GROUPED = true>
__device__ __forceinline__ EpiRaw epilogue_fetch(const LinArgs& p, uint32_t n0) {
  EpiRaw w;
  const uint32_t grp = GROUPED ? n0 / p.group_cols : 0;     // (group_cols = N for a plain launch) a block tile never straddles two groups
  // field by field: selects between kernel-argument VALUES (a struct assignment under `if` became the selection of an
  // ADDRESS in the argument segment and a second, dependent round of loads through it)
  if (GROUPED) {
#define TQ_SEL(f) w.qsel.f = grp == 0 ? p.q_out.f : (grp == 1 ? p.q_out1.f : p.q_out2.f)
    TQ_SEL(delta); TQ_SEL(zero_float); TQ_SEL(signed_flag); TQ_SEL(n_bits); TQ_SEL(symmetric); TQ_SEL(log_domain); TQ_SEL(eps);
    TQ_SEL(n_params); TQ_SEL(inner);
#undef TQ_SEL
  } else {
    w.qsel = p.q_out;
  }
  w.dx = p.x_delta[0];
  w.zx = p.x_zero_float[0];
  w.qo = load_qraw(w.qsel, 0, p.x_delta);                   // has_q == 0: reads x_delta, unused
  w.q1 = w.q2 = w.qo;
  w.t2sel = p.q_t2;
  if (WITH_TAIL) {
    if (GROUPED && p.split_out && grp == 1) w.t2sel = p.q_t2b;
    if (GROUPED && p.split_out && grp == 2) w.t2sel = p.q_t2c;
    w.q1 = load_qraw(p.q_t1, 0, p.x_delta);
    w.q2 = load_qraw(w.t2sel, 0, p.x_delta);
  }
  w.st[0] = w.st[1] = w.st[2] = w.st[3] = 0.0f;
  if (!WITH_TAIL && p.stair != nullptr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) w.st[i] = p.stair[i];
  }
  return w;
}

// every fetched value is pinned here (see qraw_arrived): call after all loads of the prologue were issued
template <bool WITH_TAIL>
__device__ __forceinline__ void epilogue_arrived(EpiRaw& w) {
  w.dx = pinned_uniform(w.dx);
  w.zx = pinned_uniform(w.zx);
  qraw_arrived(w.qo);
  if (WITH_TAIL) { qraw_arrived(w.q1); qraw_arrived(w.q2); }
  else {
#pragma unroll
    for (int i = 0; i < 4; ++i) w.st[i] = pinned_uniform(w.st[i]);
  }
}

// SEL: tq_device.h effective_scale's select form (the 128 x 128 tile kernel: see there)
template <bool WITH_TAIL, bool SEL = false>
__device__ __forceinline__ EpiCtx epilogue_finish(const LinArgs& p, const EpiRaw& w) {
  EpiCtx c;
  c.sx = w.dx < p.x_eps ? p.x_eps : w.dx;
  const int zx = (int)clamp_nanprop(rintf(w.zx), 0.0f, grid_top(p.x_n_bits));
  c.shift = 128 - zx;
  c.qo = QP{1.f, 0.f, 0.f, 0.f};
  if (p.has_q) c.qo = qp_from_raw<SEL>(w.qsel, w.qo);
  c.qf = make_qf(c.qo);
  c.fast = p.act != ACT_TANH && (!p.has_q || c.qf.ok) && p.fast_epi != 0;
  c.stair = false;
  c.st_inv_w = c.st_c0 = c.st_nbm1 = 0.0f;
  if (!WITH_TAIL && p.stair != nullptr && p.has_q && c.fast) {
    c.st_inv_w = w.st[0]; c.st_c0 = w.st[1]; c.st_nbm1 = w.st[2];
    c.stair = w.st[3] == 1.0f && c.st_nbm1 == (float)(p.stair_bins - 1);
  }
  c.qf1 = c.qf2 = c.qf;
  if (WITH_TAIL) {
    c.qf1 = make_qf(p.on_t1 ? qp_from_raw<SEL>(p.q_t1, w.q1) : QP{1.f, 0.f, 0.f, 1.f});
    c.qf2 = make_qf(p.on_t2 ? qp_from_raw<SEL>(w.t2sel, w.q2) : QP{1.f, 0.f, 0.f, 1.f});
    c.fast = c.fast && p.act == ACT_NONE && c.qf1.ok && c.qf2.ok;
  }
  return c;
}

template <bool WITH_TAIL>
__device__ __forceinline__ EpiCtx epilogue_prepare(const LinArgs& p, uint32_t n0) {
  EpiRaw w = epilogue_fetch<WITH_TAIL>(p, n0);
  epilogue_arrived<WITH_TAIL>(w);
  return epilogue_finish<WITH_TAIL>(p, w);
}

template <int NI, int MI, int YDT, bool STAGED, bool WITH_TAIL, bool FLUSH = true>
__device__ __forceinline__ void linear_epilogue(const LinArgs& p, v4i (&acc)[NI][MI], uint32_t n0, uint32_t m0, int r16,
                                                int kg, const EpiCtx& c, int8_t* stage = nullptr, const float* cst = nullptr,
                                                int cs = 2 * NI * 16, const f32x4 (*res_pre)[MI] = nullptr,
                                                const u32x2* stair_lds = nullptr) {
  if (!c.fast) return linear_epilogue_generic<NI, MI, YDT>(p, acc, n0, m0, r16, kg, c.qo, c.shift, c.sx, res_pre, FLUSH ? nullptr : stage);
  if (WITH_TAIL) {                 // separate kernel instantiation: the plain Linear keeps its register budget
#define TQ_EPI_T(Q, T) linear_epilogue_fast<NI, MI, YDT, ACT_NONE, Q, STAGED, T, FLUSH>(p, acc, n0, m0, r16, kg, c.qf, c.shift, c.sx, stage, cst, c.qf1, c.qf2, cs, res_pre)
    if (p.tail == 2) { if (p.has_q) TQ_EPI_T(true, 2); else TQ_EPI_T(false, 2); }
    else             { if (p.has_q) TQ_EPI_T(true, 1); else TQ_EPI_T(false, 1); }
#undef TQ_EPI_T
    return;
  }
  // wave-uniform dispatch: one straight-line body per (activation, quantizer) combination
#define TQ_EPI(A, Q) linear_epilogue_fast<NI, MI, YDT, A, Q, STAGED>(p, acc, n0, m0, r16, kg, c.qf, c.shift, c.sx, stage, cst, QF{}, QF{}, cs)
  if (STAGED && c.stair && stair_lds != nullptr) {
    linear_epilogue_stair<NI, MI, YDT>(p, acc, n0, m0, r16, kg, c.qf, stage, cst, cs, StairRef{stair_lds, c.st_inv_w, c.st_c0, c.st_nbm1});
    return;
  }
  if (p.has_q) {
    if (p.act == ACT_GELU)      TQ_EPI(ACT_GELU, true);
    else if (p.act == ACT_RELU) TQ_EPI(ACT_RELU, true);
    else                        TQ_EPI(ACT_NONE, true);
  } else {
    if (p.act == ACT_GELU)      TQ_EPI(ACT_GELU, false);
    else if (p.act == ACT_RELU) TQ_EPI(ACT_RELU, false);
    else                        TQ_EPI(ACT_NONE, false);
  }
#undef TQ_EPI
}

// K step of 64 bytes, one step of register prefetch (any K % 64 == 0)
template <int TN, int TM, int YDT, bool WITH_TAIL>
__global__ __launch_bounds__(kBlock) void linear_i8_k(LinArgs p) {
  constexpr int NI = TN / 16, MI = TM / 16;
  const int lane = threadIdx.x & 63;
  const uint32_t tiles_m = p.M / TM;
  const uint32_t tile = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  if (tile >= tiles_m * (p.N / TN)) return;
  const uint32_t n0 = (tile / tiles_m) * TN, m0 = (tile % tiles_m) * TM;

  const int r16 = lane & 15, kg = lane >> 4;
  const int8_t* wp = p.w + (size_t)(n0 + r16) * p.K + kg * 16;
  const int8_t* xp = p.x + (size_t)(m0 + r16) * p.K + kg * 16;

  v4i acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = v4i{0, 0, 0, 0};

  v4i fw[NI], fx[MI];
#pragma unroll
  for (int i = 0; i < NI; ++i) fw[i] = *reinterpret_cast<const v4i*>(wp + (size_t)i * 16 * p.K);
#pragma unroll
  for (int j = 0; j < MI; ++j) fx[j] = *reinterpret_cast<const v4i*>(xp + (size_t)j * 16 * p.K);

  for (uint32_t k = 64; k <= p.K; k += 64) {
    v4i nw[NI], nx[MI];
    const bool more = k < p.K;
    if (more) {
#pragma unroll
      for (int i = 0; i < NI; ++i) nw[i] = *reinterpret_cast<const v4i*>(wp + (size_t)i * 16 * p.K + k);
#pragma unroll
      for (int j = 0; j < MI; ++j) nx[j] = *reinterpret_cast<const v4i*>(xp + (size_t)j * 16 * p.K + k);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[i], fx[j], acc[i][j], 0, 0, 0);
    if (more) {
#pragma unroll
      for (int i = 0; i < NI; ++i) fw[i] = nw[i];
#pragma unroll
      for (int j = 0; j < MI; ++j) fx[j] = nx[j];
    }
  }
  linear_epilogue<NI, MI, YDT, false, WITH_TAIL>(p, acc, n0, m0, r16, kg, epilogue_prepare<WITH_TAIL>(p, n0));
}

// LDS-staged variant (the fast path).  Measured on MI355X: the LDS-free kernel above is bound by the
// vector-memory pipe -- each wave-level 16-byte load touches 16 different 128-byte lines and the L1/TA
// retires ~1 line per 4 clocks (14.6 B/clk/CU; 7.6 us for 1024x768x768, 76 us for 8192x3072x768).
// Here a block of 2 x 2 waves owns a BT x BT tile (BT = 2 * WT).  Per 128-byte K slab both operand slabs go
// straight from global memory into LDS (global_load_lds_dwordx4: no staging registers, no ds_write pass; 8 lanes
// = one 128-byte line of one row, fully coalesced).  LDS-DMA writes lane-linearly, so rows cannot be padded; the
// 16-byte chunks of a row are XOR-swizzled instead -- slot c of row r holds source chunk c ^ ((r >> 1) & 7), applied
// on the per-lane SOURCE address -- which makes every ds_read_b128 fragment read conflict-free under the real
// 4 x 16 lane grouping (rows of equal parity share a 256-byte bank row; the swizzle spreads the 8 of them over its
// 8 chunk slots).  Double-buffered: the next slab's loads are in flight while the current slab's 2 * NI * MI MFMAs
// run; one barrier per slab.  A 4-stage ring of 64-byte slabs with counted vmcnt waits was slower (tools/tuning/
// i8_glds.hip: more barriers per MFMA).  What bounds the loop was measured in round 4 (tools/tuning/i8_v4.hip, docs/history/DESIGN_rounds_1-4.md section 8):
// the operand DMA path (~56 B/clk/CU through L1/TA, one 1 KB piece per ~150 cycles and wave) and the per-slab barrier,
// NOT the LDS (ds_read_b128 peaks at 256 B/clk/CU and these fragment reads are conflict-free).  M, N % BT == 0, K % 128 == 0.
// Wait for this wave's LDS-DMA loads explicitly before the barrier that publishes them.  hipcc usually puts an
// `s_waitcnt vmcnt(0)` in front of the s_barrier of __syncthreads() when global_load_lds is in flight, but that is a
// property of its memory model pass, not a contract: after an unrelated change to the kernel prologue (epilogue
// parameters loaded before the loop) the 64 x 64 variant lost the wait inside its K loop and read half-landed slabs.
__device__ __forceinline__ void lds_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

#define TQ_GLDS16(gp, lp)                                                                         \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp),           \
                                   (__attribute__((address_space(3))) void*)(lp), 16, 0, 0)

// NS = 8 (round 5) is the same kernel with an 8-stage ring of slabs, counted waits and ONE barrier per PAIR of slabs, for
// launches of at most one block per CU (BERT-base's attention-output and second feed-forward Linear at batch 8: 192
// blocks).  With nothing else resident on the CU the double-buffered loop is a chain of exposed latencies -- per slab a
// DMA round trip when the operands are cold (inside a model forward), and barrier + fragment reads (~0.17 us, measured
// with the breakdown build: tools/tuning/i8_small_dbg.py) even when they are not: 24 slabs of K = 3072 were 8.5 us of a
// 10-14 us launch.  Here 4-6 slabs are in flight behind the pair being multiplied and the barrier count halves.
// (For the throughput-bound shapes a 4-stage ring was slower, see above: they keep NS = 2.)  K % 256 == 0.
template <int N>
__device__ __forceinline__ void lds_dma_wait_but() {                   // all but the youngest N loads of this wave have landed
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int WT, int YDT, bool WITH_TAIL, int NS = 2>
__global__ __launch_bounds__(kBlock, WT == 64 ? 2 : 4) void linear_i8_lds_k(LinArgs p) {   // 2 (128 x 128 tiles) / 4 waves per SIMD
  static_assert(NS == 2 || NS == 8, "double buffer or the 8-stage ring");
  constexpr int BT = 2 * WT, NI = WT / 16, MI = WT / 16;
  constexpr int LPW = WT / 16;                    // 1 KB load instructions per wave, operand and slab
  constexpr int OPB = BT * 128, STB = 2 * OPB;    // bytes per operand tile / per stage
  extern __shared__ __attribute__((aligned(1024))) int8_t lds_i8[];   // [2 stages][W | X][BT rows][128 B]
  prefetch_kernarg<sizeof(LinArgs)>();
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform for the compiler: scalar addressing of the epilogue parameters)
  const uint32_t tiles_m = p.M / BT;
  const int wn = (wave >> 1) * WT, wm = (wave & 1) * WT;
  const int r16 = lane & 15, kg = lane >> 4;
  // reader: k chunk c = 4 s + kg of row (16-aligned base) + r16 sits in slot c ^ ((r16 >> 1) & 7)
  const int swz = (r16 >> 1) & 7;
  const int off[2] = {r16 * 128 + ((kg ^ swz) << 4), r16 * 128 + (((4 + kg) ^ swz) << 4)};
  float* cst = reinterpret_cast<float*>(lds_i8 + NS * STB);
  u32x2* stab = reinterpret_cast<u32x2*>(lds_i8 + NS * STB + 5 * BT * 4);   // staircase entries (allocated only with p.stair)
  const uint32_t nk = p.K / 128;

  // One block per output tile.  (Round 3 tried persistent blocks working through runs of tiles -- parameters loaded
  // once per run, no per-tile dispatch: +3 % for +50 VGPRs, dropped; see docs/history/DESIGN_rounds_1-4.md section 8 for what bounds the loop.)
  {
    const uint32_t tile = blockIdx.x;
    const uint32_t n0 = (tile / tiles_m) * BT, m0 = (tile % tiles_m) * BT;
    // loader: wave w moves rows [w WT / 2, (w + 1) WT / 2) of both tiles, 8 rows per instruction
    const int8_t* wsrc[LPW];
    const int8_t* xsrc[LPW];
#pragma unroll
    for (int q = 0; q < LPW; ++q) {
      const int row = wave * (WT / 2) + q * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      wsrc[q] = p.w + (size_t)(n0 + row) * p.K + chunk * 16;
      xsrc[q] = p.x + (size_t)(m0 + row) * p.K + chunk * 16;
    }
    auto issue = [&](int stage, uint32_t k) {
#ifdef TQ_I8_DBG_BUILD
      if (p.dbg & 2) return;
#endif
      int8_t* bw = lds_i8 + stage * STB + wave * (WT / 2) * 128;
#pragma unroll
      for (int q = 0; q < LPW; ++q) {
        TQ_GLDS16(wsrc[q] + k, bw + q * 1024);
        TQ_GLDS16(xsrc[q] + k, bw + OPB + q * 1024);
      }
    };

    v4i acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = v4i{0, 0, 0, 0};

    // Everything read through pointers is REQUESTED first, as independent loads -- the quantizers' range buffers
    // (epilogue parameters; scalar loads as long as no LDS-DMA precedes them) and the per-column scale / bias / row sum
    // this thread turns into LDS constants [BT scale | BT bias | BT correction | BT NoNorm weight | BT NoNorm bias]
    // (published by the loop's first barrier) --, then the operand slabs go out (both stages / the ring's first NS - 2),
    // and only then is anything waited for: one memory round trip for the lot.  The column loads sit in front of the
    // slabs in the (in-order) vector queue, so the constants do not wait for 100 KB of operands.
    EpiRaw eraw = epilogue_fetch<WITH_TAIL>(p, n0 + wn);
    const uint32_t ncol = n0 + (tid & (BT - 1));       // threads >= BT load duplicates and do not write
    const float ld_dw = p.w_delta[p.w_n_params == 1 ? 0 : ncol], ld_b = p.bias ? p.bias[ncol] : 0.0f;
    const int ld_rs = p.w_rowsum[ncol];
    float ld_nw = 0.0f, ld_nb = 0.0f;
    if (WITH_TAIL) { ld_nw = p.nn_w[ncol]; ld_nb = p.nn_b[ncol]; }
#pragma unroll
    for (int s = 0; s < (NS == 2 ? 2 : NS - 2); ++s)
      if ((uint32_t)s < nk) issue(s, s * 128);
    epilogue_arrived<WITH_TAIL>(eraw);
    const EpiCtx ectx = epilogue_finish<WITH_TAIL, WT == 64>(p, eraw);
    if (tid < BT) {
      cst[tid] = ectx.sx * (ld_dw < p.w_eps ? p.w_eps : ld_dw);
      cst[BT + tid] = ld_b;
      reinterpret_cast<int*>(cst)[2 * BT + tid] = ld_rs * ectx.shift;
      if (WITH_TAIL) { cst[3 * BT + tid] = ld_nw; cst[4 * BT + tid] = ld_nb; }
    }
    if (!WITH_TAIL && ectx.stair) {                 // table -> LDS (L2-hot: every block reads the same <= 14 KB), published like cst
      const u32x2* gtab = reinterpret_cast<const u32x2*>(p.stair + 4);
      for (uint32_t i = tid; i < p.stair_bins; i += kBlock) stab[i] = gtab[i];
    }
    for (uint32_t kb = 0; kb < nk; ++kb) {
      if (NS == 2) {
        lds_dma_wait_all();
        __syncthreads();                              // slab kb landed (vmcnt(0) + barrier); slab kb - 1 is no longer read
        if (kb > 0 && kb + 1 < nk) issue((kb + 1) & 1, (kb + 1) * 128);     // (slab 1 went out with slab 0)
      } else if ((kb & 1) == 0) {
        // pair (kb, kb + 1): slabs up to min(kb + NS - 3, nk - 1) are out, those after kb + 1 may stay in flight (NS - 4
        // of them in the steady state).  A raw s_barrier: the compiler's __syncthreads would drain the queue
        constexpr int IPS = 2 * LPW;                  // loads per slab and wave
        const uint32_t rest = nk - 2 - kb;
        if (rest >= (uint32_t)(NS - 4)) lds_dma_wait_but<(NS - 4) * IPS>();
        else if (rest == 2) lds_dma_wait_but<2 * IPS>();
        else lds_dma_wait_but<0>();
        asm volatile("s_barrier" ::: "memory");       // the pair landed in every wave's part; the previous pair's stages are free
        if (kb + NS - 2 < nk) {
          issue((kb + NS - 2) % NS, (kb + NS - 2) * 128);
          issue((kb + NS - 1) % NS, (kb + NS - 1) * 128);
        }
      }
      const int8_t* bw = lds_i8 + (kb % NS) * STB + wn * 128;
      const int8_t* bx = lds_i8 + (kb % NS) * STB + OPB + wm * 128;
      // Fragment pipeline: the 8 LDS reads of k-step s + 1 are in flight under the 16 MFMAs of k-step s (two fragment
      // sets in registers).  Left to itself the scheduler issued one ds_read, waited for it with lgkmcnt(0), ran four
      // MFMAs, and repeated: eight exposed LDS latencies per slab, the matrix cores idle in between.
      v4i fw[2][NI], fx[2][MI];
      auto load_frags = [&](int s2) {
#pragma unroll
        for (int i = 0; i < NI; ++i) fw[s2][i] = *reinterpret_cast<const v4i*>(bw + i * 2048 + off[s2]);
#pragma unroll
        for (int j = 0; j < MI; ++j) fx[s2][j] = *reinterpret_cast<const v4i*>(bx + j * 2048 + off[s2]);
      };
      load_frags(0);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s == 0) load_frags(1);
        __builtin_amdgcn_sched_barrier(0);          // keep the reads above, the MFMAs below
#ifdef TQ_I8_DBG_BUILD
        if (p.dbg & 4) {
#pragma unroll
          for (int i = 0; i < NI; ++i) acc[i][0] = acc[i][0] + fw[s][i] + fx[s][i % MI];
          continue;
        }
#endif
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < MI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[s][i], fx[s][j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();                                // the operand stages become the waves' output staging areas
#ifdef TQ_I8_DBG_BUILD
    if (p.dbg & 1) {
      if (acc[0][0][0] == 0x7fffffff) p.y_idx[0] = 1;   // keep the accumulators alive
      return;
    }
#endif
    constexpr int kStageBytes = 32 * (WT * 4 + 16) + 32 * (WT + 16);
    static_assert(4 * kStageBytes <= 2 * STB, "output staging must fit the operand stages");   // (NS >= 2 of them)
    linear_epilogue<NI, MI, YDT, true, WITH_TAIL>(p, acc, n0 + wn, m0 + wm, r16, kg, ectx, lds_i8 + wave * kStageBytes, cst + wn,
                                                  2 * NI * 16, nullptr, stab);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// MobileBERT feed-forward block as ONE launch (reference models/quantized_mobilebert.py:330-352 on top of hijacker.py:
// 66-116): y = Q_out( Q_sum( Q_dense( lin2( Q_mid( relu( lin1(x) ) ) ) ) + residual ) * nn_w + nn_b )
//   lin1: K1 -> N1 (128 -> 512) with ReLU and the intermediate activation quantizer Q_mid (asymmetric, <= 8 bit);
//   lin2: N1 -> N2 (512 -> 128); then the NoNorm tail of tq_linear_i8_nonorm_fwd.
// At B x T = 1024 tokens these two GEMMs are 32 workgroups of a few K slabs each: pure launch + load latency (5.6 +
// 8.2 us).  Here a block owns BM = 16 or 32 token rows end to end: the [32, N1] intermediate never leaves the CU -- Q_mid's
// int8 indices go straight into LDS in the K-slab layout the second GEMM's fragments read.  Every wave works on its own
// slice of output features in BOTH GEMMs (128 of N1, then 32 of N2), so the weight slices are wave-private LDS regions
// (no barriers around them) and all three operand fetches (x rows, W1 slice, W2 slice) are in flight from the first
// instruction on; only the x tile and the intermediate are shared (2 barriers per block).  Same integer contractions,
// same element arithmetic as the separate launches: bit-identical results.
struct FfnArgs {
  const int8_t* x;          // [M, K1] indices - 128 of the block input
  const int8_t* w1;         // [N1, K1]
  const int32_t* rs1;
  const float* b1;          // or null
  const float* w1_delta;    // [1] or [N1]
  uint32_t w1_n_params;
  float w1_eps;
  const float* x_delta;     // input quantizer (per-tensor asymmetric)
  const float* x_zero_float;
  float x_eps;
  int x_n_bits;
  tq_quantizer q_mid;       // intermediate activation quantizer (= input quantizer of lin2)
  const int8_t* w2;         // [N2, N1]
  const int32_t* rs2;
  LinArgs lin2;             // second Linear + tail, as tq_linear_i8_nonorm_fwd describes it (x / w / K unused)
  uint32_t M;
};

// The three middle steps of a feed-forward block, shared by ffn_i8_k and ffn_chain_i8_k.
// GEMM 1: [BM, K1 = 128] x [N1 / 4 (own), K1]^T
template <int NI1, int MI>
__device__ __forceinline__ void ffn_gemm1(v4i (&acc1)[NI1][MI], const int8_t* bw, const int8_t* bx, const int (&off)[2]) {
#pragma unroll
  for (int i = 0; i < NI1; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc1[i][j] = v4i{0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    v4i fx[MI];
#pragma unroll
    for (int j = 0; j < MI; ++j) fx[j] = *reinterpret_cast<const v4i*>(bx + j * 2048 + off[s]);
#pragma unroll
    for (int i = 0; i < NI1; ++i) {
      const v4i fw = *reinterpret_cast<const v4i*>(bw + i * 2048 + off[s]);
#pragma unroll
      for (int j = 0; j < MI; ++j) acc1[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw, fx[j], acc1[i][j], 0, 0, 0);
    }
  }
}

// epilogue 1: scale + bias, ReLU, Q_mid -> int8(index - 128) into the intermediate's K-slab layout (slab = wave);
// two n tiles per step so that the packed quantizer chains of 2 * IT * MI register pairs overlap
// (col0: the wave's first column of GEMM 1; its tiles are chunks chunk0 .. chunk0 + NI1 - 1 of the 128-byte slab at hb)
template <int N1, int NI1, int MI, int BM>
__device__ __forceinline__ void ffn_epilogue1(const v4i (&acc1)[NI1][MI], const QP& qm, const float* c1, int8_t* hb, int col0,
                                              int chunk0, int r16, int kg) {
  const QF qf = make_qf(qm);
  const f32x2 zpb = {qm.zp, qm.zp};
  constexpr int IT = 2, NP = IT * 2 * MI;
#pragma unroll
  for (int i0 = 0; i0 < NI1; i0 += IT) {
    f32x2 v[NP], hq[NP];
#pragma unroll
    for (int ii = 0; ii < IT; ++ii) {
      const int col = col0 + (i0 + ii) * 16 + kg * 4;
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(c1 + col);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(c1 + N1 + col);
      const v4i r4 = *reinterpret_cast<const v4i*>(c1 + 2 * N1 + col);
      const f32x2 sw[2] = {f32x2{s4.x, s4.y}, f32x2{s4.z, s4.w}}, bs[2] = {f32x2{b4.x, b4.y}, f32x2{b4.z, b4.w}};
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const v4i a = acc1[i0 + ii][j];
        const f32x2 lo = {(float)(a[0] + r4.x), (float)(a[1] + r4.y)};
        const f32x2 hi = {(float)(a[2] + r4.z), (float)(a[3] + r4.w)};
        v[(ii * MI + j) * 2] = lo * sw[0] + bs[0];
        v[(ii * MI + j) * 2 + 1] = hi * sw[1] + bs[1];
      }
    }
#pragma unroll
    for (int e = 0; e < NP; ++e) v[e] = f32x2{v[e].x > 0.0f ? v[e].x : 0.0f, v[e].y > 0.0f ? v[e].y : 0.0f};
    if (qf.ok) {
      qf_round2_n<NP>(v, qf, hq);
#pragma unroll
      for (int e = 0; e < NP; ++e) hq[e] = hq[e] + zpb;
    } else {
#pragma unroll
      for (int e = 0; e < NP; ++e) hq[e] = f32x2{q_index(v[e].x, qm), q_index(v[e].y, qm)};
    }
#pragma unroll
    for (int ii = 0; ii < IT; ++ii)
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const f32x2 lo = hq[(ii * MI + j) * 2], hi = hq[(ii * MI + j) * 2 + 1];
        uint32_t w = 0;
        w = __builtin_amdgcn_cvt_pk_u8_f32(lo.x, 0, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(lo.y, 1, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(hi.x, 2, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(hi.y, 3, w) ^ 0x80808080u;
        const int m = j * 16 + r16;                    // chunk i of slab `wave`, swizzled like every operand row
        *reinterpret_cast<uint32_t*>(hb + m * 128 + (((chunk0 + i0 + ii) ^ ((m >> 1) & 7)) << 4) + kg * 4) = w;
      }
  }
}

// GEMM 2: [BM, N1] x [RW (own rows of W2), N1]^T
template <int NI2, int MI, int SL2, int RW, int BM>
__device__ __forceinline__ void ffn_gemm2(v4i (&acc2)[NI2][MI], const int8_t* w2, const int8_t* hbase, const int (&off)[2]) {
#pragma unroll
  for (int i = 0; i < NI2; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc2[i][j] = v4i{0, 0, 0, 0};
#pragma unroll
  for (int sl = 0; sl < SL2; ++sl) {
    const int8_t* bw = w2 + sl * RW * 128;
    const int8_t* bh = hbase + sl * (BM * 128);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      v4i fw[NI2], fh[MI];
#pragma unroll
      for (int i = 0; i < NI2; ++i) fw[i] = *reinterpret_cast<const v4i*>(bw + i * 2048 + off[s]);
#pragma unroll
      for (int j = 0; j < MI; ++j) fh[j] = *reinterpret_cast<const v4i*>(bh + j * 2048 + off[s]);
#pragma unroll
      for (int i = 0; i < NI2; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc2[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[i], fh[j], acc2[i][j], 0, 0, 0);
    }
  }
}

template <int K1, int N1, int N2, int YDT, int BM>
__global__ __launch_bounds__(kBlock) void ffn_i8_k(FfnArgs p) {
  static_assert(K1 == 128 && N1 == 512 && N2 == 128, "instantiated for MobileBERT's feed-forward shape");
  static_assert(BM == 16 || BM == 32, "token rows per block");
  constexpr int MI = BM / 16;
  constexpr int NI1 = N1 / 4 / 16;                  // n tiles per wave in GEMM 1 (8)
  constexpr int NI2 = N2 / 4 / 16;                  // n tiles per wave in GEMM 2 (2)
  constexpr int SL2 = N1 / 128;                     // K slabs of GEMM 2 (4)
  // LDS map (bytes): x tile | intermediate indices [SL2][BM][128] | per wave: W1 slice [N1/4][128] | per wave: W2 slice
  // [SL2][N2/4][128] | column constants of GEMM 1 [3][N1] and of GEMM 2 + tail [5][N2]
  constexpr int kX = 0, kH = kX + BM * 128, kW1 = kH + SL2 * BM * 128, kW1w = (N1 / 4) * 128;
  constexpr int kW2 = kW1 + 4 * kW1w, kW2w = SL2 * (N2 / 4) * 128, kC1 = kW2 + 4 * kW2w, kC2 = kC1 + 3 * N1 * 4;
  extern __shared__ __attribute__((aligned(1024))) int8_t lds_i8[];
  prefetch_kernarg<sizeof(FfnArgs)>();
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kg = lane >> 4;
  const uint32_t m0 = blockIdx.x * BM;

  // ---- all operand fetches first (global_load_lds: 8 rows x 128 B per instruction, XOR chunk swizzle as above)
  {
    const int row8 = lane >> 3, slot = lane & 7;
    if (wave * 8 < BM) {   // x tile: wave w brings rows [8 w, 8 w + 8)
      const int row = wave * 8 + row8;
      TQ_GLDS16(p.x + (size_t)(m0 + row) * K1 + ((slot ^ ((row >> 1) & 7)) << 4), lds_i8 + kX + wave * 1024);
    }
#pragma unroll
    for (int q = 0; q < N1 / 4 / 8; ++q) {   // own W1 slice: rows n = wave * 128 + 8 q + row8
      const int row = q * 8 + row8;
      TQ_GLDS16(p.w1 + (size_t)(wave * (N1 / 4) + row) * K1 + ((slot ^ ((row >> 1) & 7)) << 4), lds_i8 + kW1 + wave * kW1w + q * 1024);
    }
#pragma unroll
    for (int sl = 0; sl < SL2; ++sl)
#pragma unroll
      for (int q = 0; q < N2 / 4 / 8; ++q) {   // own W2 slice, slab sl: rows n = wave * 32 + 8 q + row8, k bytes [128 sl, 128 sl + 128)
        const int row = q * 8 + row8;
        TQ_GLDS16(p.w2 + (size_t)(wave * (N2 / 4) + row) * N1 + sl * 128 + ((slot ^ ((row >> 1) & 7)) << 4),
                  lds_i8 + kW2 + wave * kW2w + sl * (N2 / 4) * 128 + q * 1024);
      }
  }
  // ---- then everything read through pointers, as independent loads in flight together with the operands (measured:
  // written as load-use-load-use this prologue cost 3.4 us of dependent global round trips; round 5: the quantizers'
  // buffers, too, are fetched in one batch -- tq_device.h load_qraw): the quantizers' range buffers, the per-column
  // scales / biases / row sums this thread turns into LDS constants below, the residual values of epilogue 2
  EpiRaw eraw = epilogue_fetch<true, false>(p.lin2, wave * (N2 / 4));
  QRaw mraw = load_qraw(p.q_mid, 0, p.x_delta);
  float dx_in = p.x_delta[0], zf_in = p.x_zero_float[0];
  constexpr int C1 = N1 / kBlock;                    // GEMM 1 columns per thread (2)
  float ld_dw1[C1], ld_b1[C1];
  int ld_rs1[C1];
#pragma unroll
  for (int t = 0; t < C1; ++t) {
    const int n = tid + t * kBlock;
    ld_dw1[t] = p.w1_delta[p.w1_n_params == 1 ? 0 : n];
    ld_b1[t] = p.b1 ? p.b1[n] : 0.0f;
    ld_rs1[t] = p.rs1[n];
  }
  const LinArgs& l2 = p.lin2;
  const int n2 = tid & (N2 - 1);                     // threads >= N2 load duplicates and do not write
  const float ld_dw2 = l2.w_delta[l2.w_n_params == 1 ? 0 : n2], ld_b2 = l2.bias ? l2.bias[n2] : 0.0f;
  const int ld_rs2 = p.rs2[n2];
  const float ld_nw = l2.nn_w[n2], ld_nb = l2.nn_b[n2];
  f32x4 res_pre[NI2][MI];
#pragma unroll
  for (int i = 0; i < NI2; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j)
      res_pre[i][j] = *reinterpret_cast<const f32x4*>(l2.residual + (size_t)(m0 + j * 16 + r16) * N2 + wave * (N2 / 4) + i * 16 + kg * 4);
  epilogue_arrived<true>(eraw);
  qraw_arrived(mraw);
  dx_in = pinned_uniform(dx_in);
  zf_in = pinned_uniform(zf_in);
  const EpiCtx ectx = epilogue_finish<true>(p.lin2, eraw);
  const QP qm = qp_from_raw(p.q_mid, mraw);

  // ---- column constants (combined scale, bias, zero-point correction; NoNorm affine) -> LDS while the fetches land
  float* c1 = reinterpret_cast<float*>(lds_i8 + kC1);
  float* c2 = reinterpret_cast<float*>(lds_i8 + kC2);
  {
    const float sx = dx_in < p.x_eps ? p.x_eps : dx_in;
    const int zx = (int)clamp_nanprop(rintf(zf_in), 0.0f, grid_top(p.x_n_bits));
#pragma unroll
    for (int t = 0; t < C1; ++t) {
      const int n = tid + t * kBlock;
      c1[n] = sx * (ld_dw1[t] < p.w1_eps ? p.w1_eps : ld_dw1[t]);
      c1[N1 + n] = ld_b1[t];
      reinterpret_cast<int*>(c1)[2 * N1 + n] = ld_rs1[t] * (128 - zx);
    }
    if (tid < N2) {
      const int zm = (int)qm.zp;                       // lin2's input lives on Q_mid's grid
      c2[tid] = qm.scale * (ld_dw2 < l2.w_eps ? l2.w_eps : ld_dw2);
      c2[N2 + tid] = ld_b2;
      reinterpret_cast<int*>(c2)[2 * N2 + tid] = ld_rs2 * (128 - zm);
      c2[3 * N2 + tid] = ld_nw;
      c2[4 * N2 + tid] = ld_nb;
    }
  }
  lds_dma_wait_all();
  __syncthreads();                                   // x tile, weight slices (vmcnt(0)) and constants are in LDS

  const int swz = (r16 >> 1) & 7;
  const int off[2] = {r16 * 128 + ((kg ^ swz) << 4), r16 * 128 + (((4 + kg) ^ swz) << 4)};

  // ---- GEMM 1 -> epilogue 1 (intermediate indices into LDS) -> barrier -> GEMM 2
  v4i acc1[NI1][MI];
  ffn_gemm1<NI1, MI>(acc1, lds_i8 + kW1 + wave * kW1w, lds_i8 + kX, off);
  ffn_epilogue1<N1, NI1, MI, BM>(acc1, qm, c1, lds_i8 + kH + wave * (BM * 128), wave * (N1 / 4), 0, r16, kg);
  __syncthreads();                                   // the whole [BM, N1] intermediate is in LDS
  v4i acc2[NI2][MI];
  ffn_gemm2<NI2, MI, SL2, N2 / 4, BM>(acc2, lds_i8 + kW2 + wave * kW2w, lds_i8 + kH, off);
  // ---- epilogue 2: the NoNorm tail of the plain kernel; outputs staged through this wave's (now free) W1 region
  linear_epilogue<NI2, MI, YDT, true, true>(p.lin2, acc2, wave * (N2 / 4), m0, r16, kg, ectx, lds_i8 + kW1 + wave * kW1w,
                                            c2 + wave * (N2 / 4), N2, res_pre);
}

// ---------------------------------------------------------------------------------------------------------------------
constexpr int kMaxFfnChain = 4;
// CHAIN of feed-forward blocks as one launch (round 5).  A MobileBERT layer runs four of them back to back (reference
// models/quantized_mobilebert.py:330-352 three times under :523-526, then intermediate + output :528-529), each the
// input of the next, and NoNorm has no row statistics: everything is local to a token row, so the block of ffn_i8_k that
// owns 16 rows can take them through ALL the blocks without leaving the CU.  Per stage the same steps as ffn_i8_k (same
// integer contractions, same element arithmetic: bit-identical to n launches); between stages
//   * the stage's int8 output indices go straight into the x tile of LDS in the operand layout (every wave writes its
//     own columns; the barrier that opens the next stage publishes them), and the next residual is rebuilt from them in
//     registers: y = scale * (index - zp), the very value the tail computes from the index;
//   * the weight slices are wave-private LDS regions, so a wave refills them by itself as soon as IT is done with them:
//     W1 of the next stage right after this stage's GEMM 1, W2 right after its GEMM 2 -- each refill has about a stage
//     to land -- with counted vmcnt waits (a slice is N1 / NW / 8 LDS-DMA instructions per wave) and raw barriers;
//   * the quantizer buffers and per-column constants of ALL stages are requested at kernel start and parked in registers;
//     every wave writes the LDS constants of its OWN columns (no cross-wave hazard with their readers).
// (One kernel argument per stage: as members -- or worse, an array -- of ONE 2.7 KB argument struct, clang's private copy of
// the block survived optimisation: 2.7 KB of scratch per lane.  .x / lin2.residual are read from st0 only, lin2.y / y_idx
// belong to the last stage.)
#ifdef TQ_FFN_PROF      // tools/tuning/ffn_chain_prof.py: 8 s_memtime stamps per stage and workgroup
#define TQ_FSTAMP(k)                                                                                  \
  do {                                                                                                \
    unsigned long long t_;                                                                            \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
    if (threadIdx.x == 0 && prof) prof[(blockIdx.x * 4 + f) * 8 + (k)] = t_;                          \
  } while (0)
#define TQ_FPROF_ARG , unsigned long long* prof
#else
#define TQ_FSTAMP(k)
#define TQ_FPROF_ARG
#endif

// NW waves per workgroup (4 or 8) share the 16 rows: a wave owns N1 / NW columns of GEMM 1 and N2 / NW of GEMM 2.  A stage
// is ALU / latency bound (phase profile, tools/tuning/ffn_chain_prof.py: ~11 500 cycles with 4 waves, the LDS-DMA refills
// fully hidden), so 8 waves -- half the columns each, two waves per SIMD covering each other's stalls -- is the default.
template <int K1, int N1, int N2, int YDT, int NW>
__global__ __launch_bounds__(NW * 64) void ffn_chain_i8_k(FfnArgs st0, FfnArgs st1, FfnArgs st2, FfnArgs st3, int nf TQ_FPROF_ARG) {
  static_assert(K1 == 128 && N1 == 512 && N2 == 128, "instantiated for MobileBERT's feed-forward shape (N2 == K1: chainable)");
  static_assert(NW == 4 || NW == 8, "waves per workgroup");
  constexpr int BM = 16, MI = 1;
  constexpr int C1W = N1 / NW, C2W = N2 / NW;        // columns of GEMM 1 / GEMM 2 per wave
  constexpr int NI1 = C1W / 16, NI2 = C2W / 16, SL2 = N1 / 128;
  constexpr int ES = YDT == TQ_F32 ? 4 : 2, YP = NI2 * 16 * ES + 16, IP = NI2 * 16 + 16;   // staging pitches of the tail
  constexpr int kStage = 16 * YP + 16 * IP;          // per wave, last stage (y and indices)
  constexpr int kX = 0, kH = kX + BM * 128, kW1 = kH + SL2 * BM * 128, kW1w = C1W * 128;
  constexpr int kW2 = kW1 + NW * kW1w, kW2w = SL2 * C2W * 128, kC1 = kW2 + NW * kW2w, kC2 = kC1 + 3 * N1 * 4;
  constexpr int kS = kC2 + 5 * N2 * 4;               // the inner stages' index staging: 16 rows of pitch IP per wave
  constexpr int IW = C1W / 8;                        // LDS-DMA instructions per weight slice and wave (W1 and W2 alike)
  static_assert(IW == SL2 * (C2W / 8) && kS + NW * 16 * IP <= 160 * 1024 && kStage <= kW2w, "slices of equal size; LDS budget");
  extern __shared__ __attribute__((aligned(1024))) int8_t lds_i8[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kg = lane >> 4;
  const uint32_t m0 = blockIdx.x * BM;
  const int row8 = lane >> 3, slot = lane & 7;
  float* c1 = reinterpret_cast<float*>(lds_i8 + kC1);
  float* c2 = reinterpret_cast<float*>(lds_i8 + kC2);

  auto issue_w1 = [&](const FfnArgs& p) {     // own W1 slice: rows n = wave * C1W + 8 q + row8
#pragma unroll
    for (int q = 0; q < C1W / 8; ++q) {
      const int row = q * 8 + row8;
      TQ_GLDS16(p.w1 + (size_t)(wave * C1W + row) * K1 + ((slot ^ ((row >> 1) & 7)) << 4), lds_i8 + kW1 + wave * kW1w + q * 1024);
    }
  };
  auto issue_w2 = [&](const FfnArgs& p) {     // own W2 slice, slab sl: rows n = wave * C2W + 8 q + row8, k bytes [128 sl, 128 sl + 128)
#pragma unroll
    for (int sl = 0; sl < SL2; ++sl)
#pragma unroll
      for (int q = 0; q < C2W / 8; ++q) {
        const int row = q * 8 + row8;
        TQ_GLDS16(p.w2 + (size_t)(wave * C2W + row) * N1 + sl * 128 + ((slot ^ ((row >> 1) & 7)) << 4),
                  lds_i8 + kW2 + wave * kW2w + sl * C2W * 128 + q * 1024);
      }
  };
  // everything a stage reads through pointers, requested as independent loads and parked in registers
  struct StageRegs {
    EpiRaw eraw;
    QRaw mraw;
    float dx, zf;
    float dw1[C1W / 64], b1[C1W / 64], dw2, b2, nw, nb;
    int rs1[C1W / 64], rs2;
  };
  auto fetch = [&](const FfnArgs& p) {
    StageRegs g;
    g.eraw = epilogue_fetch<true, false>(p.lin2, wave * C2W);
    g.mraw = load_qraw(p.q_mid, 0, p.x_delta);
    g.dx = p.x_delta[0];
    g.zf = p.x_zero_float[0];
#pragma unroll
    for (int t = 0; t < C1W / 64; ++t) {        // the wave's own columns of GEMM 1
      const int n = wave * C1W + lane + t * 64;
      g.dw1[t] = p.w1_delta[p.w1_n_params == 1 ? 0 : n];
      g.b1[t] = p.b1 ? p.b1[n] : 0.0f;
      g.rs1[t] = p.rs1[n];
    }
    const LinArgs& l2 = p.lin2;
    const int n2 = wave * C2W + (lane & (C2W - 1)); // its own columns of GEMM 2 (lanes >= C2W load duplicates)
    g.dw2 = l2.w_delta[l2.w_n_params == 1 ? 0 : n2];
    g.b2 = l2.bias ? l2.bias[n2] : 0.0f;
    g.rs2 = p.rs2[n2];
    g.nw = l2.nn_w[n2];
    g.nb = l2.nn_b[n2];
    return g;
  };
  auto arrived = [&](StageRegs& g) {
    epilogue_arrived<true>(g.eraw);
    qraw_arrived(g.mraw);
    g.dx = pinned_uniform(g.dx);
    g.zf = pinned_uniform(g.zf);
  };

  // ---- operand fetches of stage 0 first (x tile, own W1 / W2 slices: in this order, the waits below count on it); then
  // the parameters of ALL stages and the residual rows.  With every ordinary load done up front, the only vector-memory
  // operations in flight during the stages are the LDS-DMA refills, and the counted waits below are exact (the compiler
  // itself waits for vmcnt(0) whenever the result of an ordinary load is needed while LDS-DMA is outstanding).
  if (wave * 8 < BM) {
    const int row = wave * 8 + row8;
    TQ_GLDS16(st0.x + (size_t)(m0 + row) * K1 + ((slot ^ ((row >> 1) & 7)) << 4), lds_i8 + kX + wave * 1024);
  }
  issue_w1(st0);
  issue_w2(st0);
  // the 2.7 KB argument block -> scalar cache while the operand fetches are on their way (tq_device.h prefetch_kernarg)
  prefetch_kernarg<4 * sizeof(FfnArgs) + sizeof(int)>();
  StageRegs g0 = fetch(st0), g1 = g0, g2 = g0, g3 = g0;
  if (nf > 1) g1 = fetch(st1);
  if (nf > 2) g2 = fetch(st2);
  if (nf > 3) g3 = fetch(st3);
  f32x4 res[NI2][MI];
#pragma unroll
  for (int i = 0; i < NI2; ++i)
    res[i][0] = *reinterpret_cast<const f32x4*>(st0.lin2.residual + (size_t)(m0 + r16) * N2 + wave * C2W + i * 16 + kg * 4);
  arrived(g0); arrived(g1); arrived(g2); arrived(g3);

  const int swz = (r16 >> 1) & 7;
  const int off[2] = {r16 * 128 + ((kg ^ swz) << 4), r16 * 128 + (((4 + kg) ^ swz) << 4)};

  // (p / pn: this stage's and the next stage's arguments)
  auto run_stage = [&](auto fc, const FfnArgs& p, const FfnArgs& pn, const StageRegs& cur) __attribute__((always_inline)) {
    constexpr int f = decltype(fc)::value;
    const LinArgs& l2 = p.lin2;
    const bool more = f + 1 < nf;
    TQ_FSTAMP(0);
    // ---- this stage's parameters -> context and LDS constants (own columns only)
    const EpiCtx ectx = epilogue_finish<true>(l2, cur.eraw);
    const QP qm = qp_from_raw(p.q_mid, cur.mraw);
    {
      const float sx = cur.dx < p.x_eps ? p.x_eps : cur.dx;
      const int zx = (int)clamp_nanprop(rintf(cur.zf), 0.0f, grid_top(p.x_n_bits));
#pragma unroll
      for (int t = 0; t < C1W / 64; ++t) {
        const int n = wave * C1W + lane + t * 64;
        c1[n] = sx * (cur.dw1[t] < p.w1_eps ? p.w1_eps : cur.dw1[t]);
        c1[N1 + n] = cur.b1[t];
        reinterpret_cast<int*>(c1)[2 * N1 + n] = cur.rs1[t] * (128 - zx);
      }
      if (lane < C2W) {
        const int n2 = wave * C2W + lane;
        const int zm = (int)qm.zp;                       // lin2's input lives on Q_mid's grid
        c2[n2] = qm.scale * (cur.dw2 < l2.w_eps ? l2.w_eps : cur.dw2);
        c2[N2 + n2] = cur.b2;
        reinterpret_cast<int*>(c2)[2 * N2 + n2] = cur.rs2 * (128 - zm);
        c2[3 * N2 + n2] = cur.nw;
        c2[4 * N2 + n2] = cur.nb;
      }
    }
    // x tile (fetched, or written by the previous stage's tails) and this wave's W1 slice; its W2 slice may still be
    // in flight.  Raw barriers: __syncthreads would drain the vector-memory queue
    TQ_FSTAMP(1);
    lds_dma_wait_but<IW>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    TQ_FSTAMP(2);
    v4i acc1[NI1][MI];
    ffn_gemm1<NI1, MI>(acc1, lds_i8 + kW1 + wave * kW1w, lds_i8 + kX, off);
    if (more) issue_w1(pn);                          // own W1 region: this wave's GEMM 1 is done with it
    TQ_FSTAMP(3);
    ffn_epilogue1<N1, NI1, MI, BM>(acc1, qm, c1, lds_i8 + kH + (wave * C1W / 128) * (BM * 128), wave * C1W, (wave * C1W % 128) / 16, r16, kg);
    TQ_FSTAMP(4);
    if (more) lds_dma_wait_but<IW>(); else lds_dma_wait_but<0>();       // this wave's W2 slice
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the whole [BM, N1] intermediate is in LDS
    TQ_FSTAMP(5);
    v4i acc2[NI2][MI];
    ffn_gemm2<NI2, MI, SL2, C2W, BM>(acc2, lds_i8 + kW2 + wave * kW2w, lds_i8 + kH, off);
    if (more) issue_w2(pn);                          // own W2 region: this wave's GEMM 2 is done with it
    TQ_FSTAMP(6);

    if (!more) {     // last stage: the NoNorm tail of the plain kernel, out to memory, staged through the wave's W2 region (done with)
      linear_epilogue<NI2, MI, YDT, true, true>(l2, acc2, wave * C2W, m0, r16, kg, ectx, lds_i8 + kW2 + wave * kW2w,
                                                c2 + wave * C2W, N2, res);
    } else {
      // the same tail with the indices kept in LDS: -> the x tile of the next stage, and y = scale * (index - zp) -- the
      // very value the tail computes from the index -- -> the next residual (registers).  (The y part of the staging
      // layout, 16 * YP bytes in front of the indices, is not touched: the area handed over starts that much earlier.)
      int8_t* istage = lds_i8 + kS + wave * (16 * IP);
      linear_epilogue<NI2, MI, YDT, true, true, false>(l2, acc2, wave * C2W, m0, r16, kg, ectx, istage - 16 * YP, c2 + wave * C2W, N2,
                                                       res);
      const float qs = ectx.qf2.scale.x, qz = ectx.qf2.zp;
#pragma unroll
      for (int i = 0; i < NI2; ++i) {
        const int col = i * 16 + kg * 4;
        const uint32_t w = *reinterpret_cast<const uint32_t*>(istage + r16 * IP + col);
        const uint32_t u = w ^ 0x80808080u;          // the four grid indices 0 .. 255
        float y4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          y4[r] = __builtin_fmaf(qs, (float)((u >> (8 * r)) & 0xffu) - qz, 0.0f);
          if (YDT != TQ_F32) {                       // bf16 storage: the next block sees the ROUNDED value, like a separate launch
            const f32x2 pr = {y4[r], 0.0f};
            const uint32_t b = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, bf16x2));
            y4[r] = __builtin_bit_cast(float, b << 16);
          }
        }
        res[i][0] = f32x4{y4[0], y4[1], y4[2], y4[3]};
        const int chunk = wave * NI2 + i;            // 16-byte chunk of the x row these 4 columns belong to
        *reinterpret_cast<uint32_t*>(lds_i8 + kX + r16 * 128 + ((chunk ^ ((r16 >> 1) & 7)) << 4) + kg * 4) = w;
      }
    }
    TQ_FSTAMP(7);
  };
  run_stage(std::integral_constant<int, 0>{}, st0, st1, g0);
  if (nf > 1) run_stage(std::integral_constant<int, 1>{}, st1, st2, g1);
  if (nf > 2) run_stage(std::integral_constant<int, 2>{}, st2, st3, g2);
  if (nf > 3) run_stage(std::integral_constant<int, 3>{}, st3, st3, g3);
}

// rowsum[n] = sum_k w[n, k]   (once per weight tensor)
__global__ __launch_bounds__(kBlock) void rowsum_i8_k(const int8_t* __restrict__ w, int32_t* __restrict__ out, uint32_t N,
                                                      uint32_t K) {
  const uint32_t n = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  int s = 0;
  for (uint32_t k = lane; k < K; k += 64) s += (int)w[(size_t)n * K + k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) out[n] = s;
}

template <int YDT, bool WITH_TAIL>
static int launch_linear_t(const LinArgs& a, hipStream_t st) {
  if (a.K % 128 == 0 && a.M % 64 == 0 && a.N % 64 == 0 && tuning("TQ_I8_LDS", 1)) {
    // 128 x 128 block tiles from 384 tiles on (1.5 per CU), else 64 x 64.  Round 6 (profiles/r06/i8_tile_ab.txt, one box, the
    // four Linear shapes of a BERT-base layer as graph replays): the threshold was 1024 tiles, which kept K = 3072 -> 768
    // on 64 x 64 tiles up to 16384 tokens -- 71.5 us against 49.1 us (27 -> 40 % of the i8 peak: with a long K the
    // epilogue is amortised and the larger tile halves the LDS traffic per MFMA); at 8192 tokens 34.3 -> 27.7 us and
    // 768 -> 768 12.0 -> 11.0 us.  Below 384 tiles the small tiles win (4096 tokens, 768 -> 768: 7.2 against 8.9 us).
    // Default-route BERT-base forward [64,128] 2.41 -> 2.28 ms, [128,128] 4.33 -> 4.07 ms; [32,128] 1.35 -> 1.37 ms.
    // (K >= 512 for the lower threshold: measured on BERT's K = 768 / 3072 only; short-K shapes keep the old rule)
    const uint64_t tiles128 = (uint64_t)(a.M / 128) * (a.N / 128);
    const bool big = a.M % 128 == 0 && a.N % 128 == 0 &&
                     (tiles128 >= 1024 || (a.K >= 512 && tiles128 >= (uint64_t)tuning("TQ_I8_BIG_MIN", 384)));
    const uint64_t grid = big ? (uint64_t)(a.M / 128) * (a.N / 128) : (uint64_t)(a.M / 64) * (a.N / 64);   // one block per tile
    // the staircase entries sit behind the per-column constants; dropped when they would cost a resident block
    // (160 KB per CU: 2 blocks of 128 x 128 tiles, 4 of 64 x 64)
    const size_t base = big ? 2 * 2 * 128 * 128 + 5 * 128 * 4 : 2 * 2 * 64 * 128 + 5 * 64 * 4;
    const size_t room = (big ? 80 * 1024 : 40 * 1024) - 512 - base;
    LinArgs b = a;
    if (b.stair != nullptr && ((size_t)b.stair_bins * 8 > room || !tuning("TQ_I8_STAIR", 1))) b.stair = nullptr;
    const size_t lds = base + (b.stair != nullptr ? (size_t)b.stair_bins * 8 : 0) + (big ? (size_t)tuning("TQ_I8_LDS_PAD", 0) : 0);
    // at most one 64 x 64 tile per CU and a long K: the 8-stage ring (up to 6 slabs = 96 KB in flight per block).  Measured
    // inside the BERT-base forward at batch 8 (profiles/r05/bert_default_route_layer_timeline.txt): K = 3072 13.9 -> 10.6 us,
    // K = 768 6.0 -> 6.2 us (six slabs: the double buffer already has a third of them in flight) -> from K = 1024 on
    constexpr int kRing = 8;
    if (!big && b.stair == nullptr && a.K % 256 == 0 && a.K >= (uint32_t)tuning("TQ_I8_RING_MIN_K", 1024) &&
        grid <= (uint64_t)tuning("TQ_I8_RING_MAX_GRID", 256)) {
      auto k = linear_i8_lds_k<32, YDT, WITH_TAIL, kRing>;
      const size_t ring_lds = (size_t)kRing * 2 * 64 * 128 + 5 * 64 * 4;
      static bool attr_set[64] = {};                // per instantiation and device; benign if two threads both set it
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
      if (dev < 0 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring_lds) != hipSuccess)
          return set_error(TQ_ELAUNCH, "linear_i8_lds_k (ring): cannot reserve %zu bytes of LDS", ring_lds);
        if (dev >= 0) attr_set[dev] = true;
      }
      hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kBlock), ring_lds, st, b);
      return check_launch("linear_i8_lds_k (ring)");
    }
    // at most one 64 x 64 tile per CU (each wave alone on its SIMD) and a short K: 32 x 32 tiles, four times the workgroups
    // -- several resident per CU cover each other's latencies.  Measured in the model forwards at batch 8: BERT-base's
    // attention-output Linear 5.9 -> 5.4 us, MobileBERT W4A4 (every Linear is this small) 1.383 -> 1.305 ms; for K >= 1024
    // the ring above stays ahead (BERT-base 0.699 vs 0.714 ms with 32 x 32 tiles there too)
    if (!big && b.stair == nullptr && grid <= (uint64_t)tuning("TQ_I8_SMALL_MAX_GRID", 256)) {
      const uint64_t grid32 = (uint64_t)(a.M / 32) * (a.N / 32);
      const size_t lds32 = 2 * 2 * 32 * 128 + 5 * 32 * 4;
      hipLaunchKernelGGL((linear_i8_lds_k<16, YDT, WITH_TAIL>), dim3((unsigned)grid32), dim3(kBlock), lds32, st, b);
      return check_launch("linear_i8_lds_k (32 x 32 tiles)");
    }
    if (big) hipLaunchKernelGGL((linear_i8_lds_k<64, YDT, WITH_TAIL>), dim3((unsigned)grid), dim3(kBlock), lds, st, b);
    else     hipLaunchKernelGGL((linear_i8_lds_k<32, YDT, WITH_TAIL>), dim3((unsigned)grid), dim3(kBlock), lds, st, b);
    return check_launch("linear_i8_lds_k");
  }
  // odd shapes (M, N % 32 == 0, K % 64 == 0): LDS-free kernel, 32 x 32 wave tiles
  const unsigned grid = (unsigned)ceil_div((uint64_t)(a.M / 32) * (a.N / 32), kBlock / kWave);
  hipLaunchKernelGGL((linear_i8_k<32, 32, YDT, WITH_TAIL>), dim3(grid), dim3(kBlock), 0, st, a);
  return check_launch("linear_i8_k");
}

template <int YDT>
static int launch_linear(LinArgs a, hipStream_t st) {
  a.fast_epi = tuning("TQ_I8_FAST_EPI", 1);
  a.dbg = tuning("TQ_I8_DBG", 0);
  return a.tail ? launch_linear_t<YDT, true>(a, st) : launch_linear_t<YDT, false>(a, st);
}

}  // namespace tq

using namespace tq;

extern "C" int tq_rowsum_i8(const int8_t* w_idx, int32_t* rowsum, uint64_t N, uint64_t K, tq_stream_t stream) {
  TQ_REQUIRE(w_idx && rowsum && N >= 1 && K >= 1, "tq_rowsum_i8: bad argument");
  hipLaunchKernelGGL(rowsum_i8_k, dim3((unsigned)ceil_div(N, kBlock / kWave)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), w_idx, rowsum, (uint32_t)N, (uint32_t)K);
  return check_launch("rowsum_i8_k");
}

extern "C" int tq_linear_i8_grouped_fwd(const int8_t* x_idx, const int8_t* w_idx, const int32_t* w_rowsum,
                                        const float* bias, void* y, int8_t* y_idx, int y_dtype, uint64_t M, uint64_t N,
                                        uint64_t K, const float* x_delta, const float* x_zero_float, int x_n_bits,
                                        float x_eps, const float* w_delta, float w_eps, int activation,
                                        uint64_t n_groups, const tq_quantizer* const* q_out, tq_stream_t stream) {
  if (M == 0 || N == 0) return TQ_OK;
  TQ_REQUIRE(x_idx && w_idx && w_rowsum && x_delta && x_zero_float && w_delta && q_out, "tq_linear_i8_grouped_fwd: NULL pointer");
  TQ_REQUIRE(y != nullptr || y_idx != nullptr, "tq_linear_i8_grouped_fwd: no output requested");
  TQ_REQUIRE(y_dtype == TQ_F32 || y_dtype == TQ_BF16, "tq_linear_i8_grouped_fwd: y dtype must be fp32 or bf16");
  TQ_REQUIRE(n_groups >= 1 && n_groups <= 3 && N % n_groups == 0 && (N / n_groups) % 64 == 0,
             "tq_linear_i8_grouped_fwd: 1..3 groups of a multiple of 64 output features");
  TQ_REQUIRE(M % 64 == 0 && K % 128 == 0 && K <= 16384 && M < (1u << 31) && N < (1u << 31),
             "tq_linear_i8_grouped_fwd: unsupported shape M=%llu N=%llu K=%llu (M %% 64, K %% 128)", (unsigned long long)M,
             (unsigned long long)N, (unsigned long long)K);
  TQ_REQUIRE(x_n_bits >= 1 && x_n_bits <= 8, "tq_linear_i8_grouped_fwd: input quantizer must have <= 8 bits");
  TQ_REQUIRE(activation >= ACT_NONE && activation <= ACT_TANH, "tq_linear_i8_grouped_fwd: unknown activation %d", activation);
  TQ_REQUIRE(aligned16(x_idx) && aligned16(w_idx) && (y == nullptr || aligned16(y)), "tq_linear_i8_grouped_fwd: 16-byte alignment required");
  LinArgs a{};
  a.x = x_idx; a.w = w_idx; a.w_rowsum = w_rowsum; a.bias = bias; a.y = y; a.y_idx = y_idx;
  a.M = (uint32_t)M; a.N = (uint32_t)N; a.K = (uint32_t)K;
  a.x_delta = x_delta; a.x_zero_float = x_zero_float; a.x_eps = x_eps; a.x_n_bits = x_n_bits;
  a.w_delta = w_delta; a.w_n_params = (uint32_t)N; a.w_eps = w_eps; a.act = activation;
  a.has_q = 1;
  a.group_cols = (uint32_t)(N / n_groups);
  tq_quantizer* slots[3] = {&a.q_out, &a.q_out1, &a.q_out2};
  for (uint64_t g = 0; g < n_groups; ++g) {
    TQ_REQUIRE(q_out[g] != nullptr, "tq_linear_i8_grouped_fwd: every group needs an output quantizer");
    if (int e = check_quantizer(q_out[g], M * N, "tq_linear_i8_grouped_fwd")) return e;
    TQ_REQUIRE(q_out[g]->n_params == 1, "tq_linear_i8_grouped_fwd: per-tensor output quantizers only");
    TQ_REQUIRE(y_idx == nullptr || (!q_out[g]->symmetric && q_out[g]->n_bits <= 8),
               "tq_linear_i8_grouped_fwd: y_idx needs asymmetric <= 8-bit output quantizers");
    *slots[g] = *q_out[g];
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  return y_dtype == TQ_F32 ? launch_linear<TQ_F32>(a, st) : launch_linear<TQ_BF16>(a, st);
}

extern "C" int tq_linear_i8_fwd(const int8_t* x_idx, const int8_t* w_idx, const int32_t* w_rowsum, const float* bias,
                                void* y, int8_t* y_idx, int y_dtype, uint64_t M, uint64_t N, uint64_t K, const float* x_delta,
                                const float* x_zero_float, int x_n_bits, float x_eps, const float* w_delta,
                                uint64_t w_n_params, float w_eps, int activation, const tq_quantizer* q_out,
                                tq_stream_t stream) {
  return tq_linear_i8_stair_fwd(x_idx, w_idx, w_rowsum, bias, y, y_idx, y_dtype, M, N, K, x_delta, x_zero_float, x_n_bits, x_eps,
                                w_delta, w_n_params, w_eps, activation, q_out, nullptr, 0, stream);
}

extern "C" int tq_linear_i8_stair_fwd(const int8_t* x_idx, const int8_t* w_idx, const int32_t* w_rowsum, const float* bias,
                                      void* y, int8_t* y_idx, int y_dtype, uint64_t M, uint64_t N, uint64_t K,
                                      const float* x_delta, const float* x_zero_float, int x_n_bits, float x_eps,
                                      const float* w_delta, uint64_t w_n_params, float w_eps, int activation,
                                      const tq_quantizer* q_out, const void* act_stair, uint32_t stair_bins,
                                      tq_stream_t stream) {
  if (M == 0 || N == 0) return TQ_OK;
  TQ_REQUIRE(act_stair == nullptr || (q_out != nullptr && stair_bins >= 64 && stair_bins <= 2048 && aligned16(act_stair)),
             "tq_linear_i8_stair_fwd: the staircase needs an output quantizer, 64..2048 bins and 16-byte alignment");
  TQ_REQUIRE(x_idx && w_idx && w_rowsum && (y || y_idx) && x_delta && x_zero_float && w_delta, "tq_linear_i8_fwd: NULL pointer");
  TQ_REQUIRE(y_dtype == TQ_F32 || y_dtype == TQ_BF16, "tq_linear_i8_fwd: y dtype must be fp32 or bf16");
  TQ_REQUIRE(M % 32 == 0 && N % 32 == 0 && K % 64 == 0 && K >= 64 && K <= 16384 && M < (1u << 31) && N < (1u << 31),
             "tq_linear_i8_fwd: unsupported shape M=%llu N=%llu K=%llu (M,N %% 32, K %% 64)", (unsigned long long)M,
             (unsigned long long)N, (unsigned long long)K);
  TQ_REQUIRE(x_n_bits >= 1 && x_n_bits <= 8, "tq_linear_i8_fwd: input quantizer must have <= 8 bits");
  TQ_REQUIRE(w_n_params == 1 || w_n_params == N, "tq_linear_i8_fwd: weight scales must be per-tensor or per-output-channel");
  TQ_REQUIRE(activation >= ACT_NONE && activation <= ACT_TANH, "tq_linear_i8_fwd: unknown activation %d", activation);
  TQ_REQUIRE(aligned16(x_idx) && aligned16(w_idx) && (y == nullptr || aligned16(y)), "tq_linear_i8_fwd: 16-byte alignment required");
  LinArgs a{};
  a.x = x_idx; a.w = w_idx; a.w_rowsum = w_rowsum; a.bias = bias; a.y = y; a.y_idx = y_idx;
  a.M = (uint32_t)M; a.N = (uint32_t)N; a.K = (uint32_t)K;
  a.x_delta = x_delta; a.x_zero_float = x_zero_float; a.x_eps = x_eps; a.x_n_bits = x_n_bits;
  a.w_delta = w_delta; a.w_n_params = (uint32_t)w_n_params; a.w_eps = w_eps; a.act = activation;
  a.has_q = q_out != nullptr;
  a.group_cols = (uint32_t)N;
  TQ_REQUIRE(y_idx == nullptr || (q_out != nullptr && !q_out->symmetric && q_out->n_bits <= 8),
             "tq_linear_i8_fwd: y_idx needs an asymmetric <= 8-bit output quantizer");
  if (q_out) {
    if (int e = check_quantizer(q_out, M * N, "tq_linear_i8_fwd")) return e;
    TQ_REQUIRE(q_out->n_params == 1, "tq_linear_i8_fwd: per-tensor output quantizer only");
    a.q_out = *q_out;
  }
  a.stair = static_cast<const float*>(act_stair);
  a.stair_bins = stair_bins;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return y_dtype == TQ_F32 ? launch_linear<TQ_F32>(a, st) : launch_linear<TQ_BF16>(a, st);
}

// Linear -> (+ residual) -> NoNorm -> quantizers as ONE launch (MobileBERT's bottlenecks and its four residual tails per
// layer; reference models/quantized_mobilebert.py:58-72 with :287-304, :330-352):
//   residual == NULL:  y = Q_out( Q_dense(lin) * nn_weight + nn_bias )
//   else:              y = Q_out( Q_sum( Q_dense(lin) + residual ) * nn_weight + nn_bias ),   lin = F.linear(Q_x(x), Q_w(W), b)
// Each quantizer may be NULL (identity).  Same integer contraction and the same element arithmetic as
// tq_linear_i8_fwd followed by tq_residual_nonorm_quant_fwd / tq_affine_fake_quant_fwd: bit-identical results.
extern "C" int tq_linear_i8_nonorm_fwd(const int8_t* x_idx, const int8_t* w_idx, const int32_t* w_rowsum, const float* bias,
                                       const float* residual, const float* nn_weight, const float* nn_bias, void* y,
                                       int8_t* y_idx, int y_dtype, uint64_t M, uint64_t N, uint64_t K, const float* x_delta,
                                       const float* x_zero_float, int x_n_bits, float x_eps, const float* w_delta,
                                       uint64_t w_n_params, float w_eps, const tq_quantizer* q_dense,
                                       const tq_quantizer* q_sum, const tq_quantizer* q_out, tq_stream_t stream) {
  if (M == 0 || N == 0) return TQ_OK;
  TQ_REQUIRE(x_idx && w_idx && w_rowsum && y && x_delta && x_zero_float && w_delta && nn_weight && nn_bias,
             "tq_linear_i8_nonorm_fwd: NULL pointer");
  TQ_REQUIRE(y_dtype == TQ_F32 || y_dtype == TQ_BF16, "tq_linear_i8_nonorm_fwd: y dtype must be fp32 or bf16");
  TQ_REQUIRE(M % 32 == 0 && N % 32 == 0 && K % 64 == 0 && K >= 64 && K <= 16384 && M < (1u << 31) && N < (1u << 31),
             "tq_linear_i8_nonorm_fwd: unsupported shape M=%llu N=%llu K=%llu (M,N %% 32, K %% 64)", (unsigned long long)M,
             (unsigned long long)N, (unsigned long long)K);
  TQ_REQUIRE(x_n_bits >= 1 && x_n_bits <= 8, "tq_linear_i8_nonorm_fwd: input quantizer must have <= 8 bits");
  TQ_REQUIRE(w_n_params == 1 || w_n_params == N, "tq_linear_i8_nonorm_fwd: weight scales must be per-tensor or per-output-channel");
  TQ_REQUIRE(aligned16(x_idx) && aligned16(w_idx) && aligned16(y) && (residual == nullptr || aligned16(residual)),
             "tq_linear_i8_nonorm_fwd: 16-byte alignment required");
  TQ_REQUIRE(q_sum == nullptr || residual != nullptr, "tq_linear_i8_nonorm_fwd: q_sum without a residual");
  TQ_REQUIRE(y_idx == nullptr || (q_out != nullptr && !q_out->symmetric && q_out->n_bits <= 8),
             "tq_linear_i8_nonorm_fwd: y_idx needs an asymmetric <= 8-bit output quantizer");
  LinArgs a{};
  a.x = x_idx; a.w = w_idx; a.w_rowsum = w_rowsum; a.bias = bias; a.y = y; a.y_idx = y_idx;
  a.M = (uint32_t)M; a.N = (uint32_t)N; a.K = (uint32_t)K;
  a.x_delta = x_delta; a.x_zero_float = x_zero_float; a.x_eps = x_eps; a.x_n_bits = x_n_bits;
  a.w_delta = w_delta; a.w_n_params = (uint32_t)w_n_params; a.w_eps = w_eps; a.act = ACT_NONE;
  a.group_cols = (uint32_t)N;
  a.tail = residual != nullptr ? 2 : 1;
  a.residual = residual; a.nn_w = nn_weight; a.nn_b = nn_bias;
  const tq_quantizer* qs[3] = {q_dense, q_sum, q_out};
  for (const tq_quantizer* q : qs)
    if (q != nullptr) {
      if (int e = check_quantizer(q, M * N, "tq_linear_i8_nonorm_fwd")) return e;
      TQ_REQUIRE(q->n_params == 1, "tq_linear_i8_nonorm_fwd: per-tensor quantizers only");
    }
  a.has_q = q_dense != nullptr;
  if (q_dense) a.q_out = *q_dense;
  a.on_t1 = q_sum != nullptr;
  if (q_sum) a.q_t1 = *q_sum;
  a.on_t2 = q_out != nullptr;
  if (q_out) a.q_t2 = *q_out;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return y_dtype == TQ_F32 ? launch_linear<TQ_F32>(a, st) : launch_linear<TQ_BF16>(a, st);
}

// MobileBERT feed-forward block (intermediate Linear + ReLU + quantizer, output Linear, residual NoNorm tail) as one
// launch: see ffn_i8_k.  Shapes: (K1, N1, N2) = (128, 512, 128), M % 32 == 0; anything else is TQ_EINVAL and the caller
// runs tq_linear_i8_fwd + tq_linear_i8_nonorm_fwd (the results are bit-identical either way).
extern "C" int tq_linear_i8_nonorm_grouped_fwd(const int8_t* x_idx, const int8_t* w_idx, const int32_t* w_rowsum,
                                               const float* bias, const float* nn_weight, const float* nn_bias, void* y,
                                               int8_t* y_idx, int y_dtype, uint64_t M, uint64_t N, uint64_t K,
                                               const float* x_delta, const float* x_zero_float, int x_n_bits, float x_eps,
                                               const float* w_delta, float w_eps, uint64_t n_groups,
                                               const tq_quantizer* const* q_dense, const tq_quantizer* const* q_out,
                                               tq_stream_t stream) {
  if (M == 0 || N == 0) return TQ_OK;
  TQ_REQUIRE(x_idx && w_idx && w_rowsum && y && x_delta && x_zero_float && w_delta && nn_weight && nn_bias,
             "tq_linear_i8_nonorm_grouped_fwd: NULL pointer");
  TQ_REQUIRE(y_dtype == TQ_F32 || y_dtype == TQ_BF16, "tq_linear_i8_nonorm_grouped_fwd: y dtype must be fp32 or bf16");
  TQ_REQUIRE((n_groups == 2 || n_groups == 3) && N % n_groups == 0 && (N / n_groups) % 64 == 0,
             "tq_linear_i8_nonorm_grouped_fwd: 2 or 3 groups of a multiple of 64 output features");
  const int G = (int)n_groups;
  TQ_REQUIRE(M % 64 == 0 && K % 128 == 0 && K <= 16384 && M < (1u << 31) && N < (1u << 31),
             "tq_linear_i8_nonorm_grouped_fwd: unsupported shape M=%llu N=%llu K=%llu (M %% 64, K %% 128)", (unsigned long long)M,
             (unsigned long long)N, (unsigned long long)K);
  TQ_REQUIRE(x_n_bits >= 1 && x_n_bits <= 8, "tq_linear_i8_nonorm_grouped_fwd: input quantizer must have <= 8 bits");
  TQ_REQUIRE(aligned16(x_idx) && aligned16(w_idx) && aligned16(y), "tq_linear_i8_nonorm_grouped_fwd: 16-byte alignment required");
  LinArgs a{};
  a.x = x_idx; a.w = w_idx; a.w_rowsum = w_rowsum; a.bias = bias; a.y = y; a.y_idx = y_idx;
  a.M = (uint32_t)M; a.N = (uint32_t)N; a.K = (uint32_t)K;
  a.x_delta = x_delta; a.x_zero_float = x_zero_float; a.x_eps = x_eps; a.x_n_bits = x_n_bits;
  a.w_delta = w_delta; a.w_n_params = (uint32_t)N; a.w_eps = w_eps; a.act = ACT_NONE;
  a.group_cols = (uint32_t)(N / n_groups);
  a.split_out = 1;
  a.tail = 1;
  a.nn_w = nn_weight; a.nn_b = nn_bias;
  const bool has_d = q_dense != nullptr && q_dense[0] != nullptr, has_o = q_out != nullptr && q_out[0] != nullptr;
  for (int g = 1; g < G; ++g) {
    TQ_REQUIRE(has_d == (q_dense != nullptr && q_dense[g] != nullptr), "tq_linear_i8_nonorm_grouped_fwd: all groups or none need a dense-output quantizer");
    TQ_REQUIRE(has_o == (q_out != nullptr && q_out[g] != nullptr), "tq_linear_i8_nonorm_grouped_fwd: all groups or none need an output quantizer");
  }
  TQ_REQUIRE(y_idx == nullptr || has_o, "tq_linear_i8_nonorm_grouped_fwd: y_idx needs output quantizers");
  for (int g = 0; g < G; ++g) {
    const tq_quantizer* qs[2] = {has_d ? q_dense[g] : nullptr, has_o ? q_out[g] : nullptr};
    for (const tq_quantizer* q : qs)
      if (q != nullptr) {
        if (int e = check_quantizer(q, M * N, "tq_linear_i8_nonorm_grouped_fwd")) return e;
        TQ_REQUIRE(q->n_params == 1, "tq_linear_i8_nonorm_grouped_fwd: per-tensor quantizers only");
      }
    TQ_REQUIRE(y_idx == nullptr || (!q_out[g]->symmetric && q_out[g]->n_bits <= 8),
               "tq_linear_i8_nonorm_grouped_fwd: y_idx needs asymmetric <= 8-bit output quantizers");
  }
  a.has_q = has_d;
  if (has_d) { a.q_out = *q_dense[0]; a.q_out1 = *q_dense[1]; if (G == 3) a.q_out2 = *q_dense[2]; }
  a.on_t2 = has_o;
  if (has_o) { a.q_t2 = *q_out[0]; a.q_t2b = *q_out[1]; if (G == 3) a.q_t2c = *q_out[2]; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  return y_dtype == TQ_F32 ? launch_linear<TQ_F32>(a, st) : launch_linear<TQ_BF16>(a, st);
}

extern "C" int tq_ffn_i8_nonorm_fwd(const int8_t* x_idx, const float* x_delta, const float* x_zero_float, int x_n_bits, float x_eps,
                                    const int8_t* w1_idx, const int32_t* w1_rowsum, const float* bias1, const float* w1_delta,
                                    uint64_t w1_n_params, float w1_eps, const tq_quantizer* q_mid, const int8_t* w2_idx,
                                    const int32_t* w2_rowsum, const float* bias2, const float* w2_delta, uint64_t w2_n_params,
                                    float w2_eps, const float* residual, const float* nn_weight, const float* nn_bias,
                                    const tq_quantizer* q_dense, const tq_quantizer* q_sum, const tq_quantizer* q_out, void* y,
                                    int8_t* y_idx, int y_dtype, uint64_t M, uint64_t K1, uint64_t N1, uint64_t N2,
                                    tq_stream_t stream) {
  if (M == 0) return TQ_OK;
  TQ_REQUIRE(x_idx && x_delta && x_zero_float && w1_idx && w1_rowsum && w1_delta && q_mid && w2_idx && w2_rowsum && w2_delta &&
             residual && nn_weight && nn_bias && y, "tq_ffn_i8_nonorm_fwd: NULL pointer");
  TQ_REQUIRE(K1 == 128 && N1 == 512 && N2 == 128, "tq_ffn_i8_nonorm_fwd: only the (128, 512, 128) feed-forward shape is built");
  TQ_REQUIRE(M % 32 == 0 && M < (1u << 31), "tq_ffn_i8_nonorm_fwd: M must be a multiple of 32");
  TQ_REQUIRE(y_dtype == TQ_F32 || y_dtype == TQ_BF16, "tq_ffn_i8_nonorm_fwd: y dtype must be fp32 or bf16");
  TQ_REQUIRE(x_n_bits >= 1 && x_n_bits <= 8, "tq_ffn_i8_nonorm_fwd: input quantizer must have <= 8 bits");
  TQ_REQUIRE((w1_n_params == 1 || w1_n_params == N1) && (w2_n_params == 1 || w2_n_params == N2),
             "tq_ffn_i8_nonorm_fwd: weight scales must be per-tensor or per-output-channel");
  TQ_REQUIRE(aligned16(x_idx) && aligned16(w1_idx) && aligned16(w2_idx) && aligned16(y) && aligned16(residual),
             "tq_ffn_i8_nonorm_fwd: 16-byte alignment required");
  if (int e = check_quantizer(q_mid, M * N1, "tq_ffn_i8_nonorm_fwd")) return e;
  TQ_REQUIRE(q_mid->n_params == 1 && !q_mid->symmetric && q_mid->n_bits <= 8 && !q_mid->log_domain,
             "tq_ffn_i8_nonorm_fwd: the intermediate quantizer must be per-tensor, asymmetric, linear, <= 8 bit");
  TQ_REQUIRE(y_idx == nullptr || (q_out != nullptr && !q_out->symmetric && q_out->n_bits <= 8),
             "tq_ffn_i8_nonorm_fwd: y_idx needs an asymmetric <= 8-bit output quantizer");
  FfnArgs f{};
  f.x = x_idx; f.w1 = w1_idx; f.rs1 = w1_rowsum; f.b1 = bias1; f.w1_delta = w1_delta; f.w1_n_params = (uint32_t)w1_n_params;
  f.w1_eps = w1_eps; f.x_delta = x_delta; f.x_zero_float = x_zero_float; f.x_eps = x_eps; f.x_n_bits = x_n_bits;
  f.q_mid = *q_mid; f.w2 = w2_idx; f.rs2 = w2_rowsum; f.M = (uint32_t)M;
  LinArgs& a = f.lin2;
  a.w_rowsum = w2_rowsum; a.bias = bias2; a.y = y; a.y_idx = y_idx;
  a.M = (uint32_t)M; a.N = (uint32_t)N2; a.K = (uint32_t)N1;
  a.x_delta = q_mid->delta; a.x_zero_float = q_mid->zero_float; a.x_eps = q_mid->eps; a.x_n_bits = q_mid->n_bits;
  a.w_delta = w2_delta; a.w_n_params = (uint32_t)w2_n_params; a.w_eps = w2_eps; a.act = ACT_NONE;
  a.group_cols = (uint32_t)N2; a.fast_epi = tuning("TQ_I8_FAST_EPI", 1);
  a.tail = 2; a.residual = residual; a.nn_w = nn_weight; a.nn_b = nn_bias;
  const tq_quantizer* qs[3] = {q_dense, q_sum, q_out};
  for (const tq_quantizer* q : qs)
    if (q != nullptr) {
      if (int e = check_quantizer(q, M * N2, "tq_ffn_i8_nonorm_fwd")) return e;
      TQ_REQUIRE(q->n_params == 1, "tq_ffn_i8_nonorm_fwd: per-tensor quantizers only");
    }
  a.has_q = q_dense != nullptr;
  if (q_dense) a.q_out = *q_dense;
  a.on_t1 = q_sum != nullptr;
  if (q_sum) a.q_t1 = *q_sum;
  a.on_t2 = q_out != nullptr;
  if (q_out) a.q_t2 = *q_out;
  // 16 token rows per block while that still leaves every CU at most one block (more, shorter blocks), else 32
  const int bm = tuning("TQ_FFN_BM", M / 16 <= 256 ? 16 : 32);
  TQ_REQUIRE(bm == 16 || bm == 32, "TQ_FFN_BM must be 16 or 32");
  const size_t lds = (size_t)bm * 128 + 4 * (size_t)bm * 128 + 4 * 128 * 128 + 4 * 4 * 32 * 128 + 3 * 512 * 4 + 5 * 128 * 4;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)(M / bm));
#define TQ_FFN(DT, B) hipLaunchKernelGGL((ffn_i8_k<128, 512, 128, DT, B>), grid, dim3(kBlock), lds, st, f)
  if (y_dtype == TQ_F32) { if (bm == 16) TQ_FFN(TQ_F32, 16); else TQ_FFN(TQ_F32, 32); }
  else                   { if (bm == 16) TQ_FFN(TQ_BF16, 16); else TQ_FFN(TQ_BF16, 32); }
#undef TQ_FFN
  return check_launch("ffn_i8_k");
}

extern "C" int tq_ffn_chain_i8_nonorm_fwd(const int8_t* x_idx, const float* x_delta, const float* x_zero_float, int x_n_bits,
                                          float x_eps, const float* residual, const tq_ffn_stage* stages, uint64_t n_stages,
                                          void* y, int8_t* y_idx, int y_dtype, uint64_t M, uint64_t K1, uint64_t N1, uint64_t N2,
                                          tq_stream_t stream) {
  if (M == 0) return TQ_OK;
  TQ_REQUIRE(x_idx && x_delta && x_zero_float && residual && stages && y, "tq_ffn_chain_i8_nonorm_fwd: NULL pointer");
  TQ_REQUIRE(n_stages >= 2 && n_stages <= (uint64_t)kMaxFfnChain, "tq_ffn_chain_i8_nonorm_fwd: 2..%d feed-forward blocks (one: tq_ffn_i8_nonorm_fwd)", kMaxFfnChain);
  TQ_REQUIRE(K1 == 128 && N1 == 512 && N2 == 128, "tq_ffn_chain_i8_nonorm_fwd: only the (128, 512, 128) feed-forward shape is built");
  TQ_REQUIRE(M % 16 == 0 && M < (1u << 31), "tq_ffn_chain_i8_nonorm_fwd: M must be a multiple of 16");
  TQ_REQUIRE(y_dtype == TQ_F32 || y_dtype == TQ_BF16, "tq_ffn_chain_i8_nonorm_fwd: y dtype must be fp32 or bf16");
  TQ_REQUIRE(x_n_bits >= 1 && x_n_bits <= 8, "tq_ffn_chain_i8_nonorm_fwd: input quantizer must have <= 8 bits");
  TQ_REQUIRE(aligned16(x_idx) && aligned16(y) && aligned16(residual), "tq_ffn_chain_i8_nonorm_fwd: 16-byte alignment required");
  FfnArgs c[kMaxFfnChain] = {};
  const int fast_epi = tuning("TQ_I8_FAST_EPI", 1);
  for (uint64_t s = 0; s < n_stages; ++s) {
    const tq_ffn_stage& g = stages[s];
    const bool last = s + 1 == n_stages;
    TQ_REQUIRE(g.w1_idx && g.w1_rowsum && g.w1_delta && g.q_mid && g.w2_idx && g.w2_rowsum && g.w2_delta && g.nn_weight && g.nn_bias,
               "tq_ffn_chain_i8_nonorm_fwd: NULL pointer in stage %llu", (unsigned long long)s);
    TQ_REQUIRE((g.w1_n_params == 1 || g.w1_n_params == N1) && (g.w2_n_params == 1 || g.w2_n_params == N2),
               "tq_ffn_chain_i8_nonorm_fwd: weight scales must be per-tensor or per-output-channel");
    TQ_REQUIRE(aligned16(g.w1_idx) && aligned16(g.w2_idx), "tq_ffn_chain_i8_nonorm_fwd: 16-byte alignment required");
    if (int e = check_quantizer(g.q_mid, M * N1, "tq_ffn_chain_i8_nonorm_fwd")) return e;
    TQ_REQUIRE(g.q_mid->n_params == 1 && !g.q_mid->symmetric && g.q_mid->n_bits <= 8 && !g.q_mid->log_domain,
               "tq_ffn_chain_i8_nonorm_fwd: the intermediate quantizer must be per-tensor, asymmetric, linear, <= 8 bit");
    const tq_quantizer* qs[3] = {g.q_dense, g.q_sum, g.q_out};
    for (const tq_quantizer* q : qs)
      if (q != nullptr) {
        if (int e = check_quantizer(q, M * N2, "tq_ffn_chain_i8_nonorm_fwd")) return e;
        TQ_REQUIRE(q->n_params == 1, "tq_ffn_chain_i8_nonorm_fwd: per-tensor quantizers only");
      }
    // a block's output feeds the next block's integer GEMM: it must live on an asymmetric, linear <= 8-bit grid
    TQ_REQUIRE((last && y_idx == nullptr) || (g.q_out != nullptr && !g.q_out->symmetric && g.q_out->n_bits <= 8 && !g.q_out->log_domain),
               "tq_ffn_chain_i8_nonorm_fwd: stage %llu needs an asymmetric, linear <= 8-bit output quantizer", (unsigned long long)s);
    FfnArgs& f = c[s];
    f.x = x_idx; f.w1 = g.w1_idx; f.rs1 = g.w1_rowsum; f.b1 = g.bias1; f.w1_delta = g.w1_delta;
    f.w1_n_params = (uint32_t)g.w1_n_params; f.w1_eps = g.w1_eps;
    if (s == 0) { f.x_delta = x_delta; f.x_zero_float = x_zero_float; f.x_eps = x_eps; f.x_n_bits = x_n_bits; }
    else {
      const tq_quantizer* in = stages[s - 1].q_out;
      f.x_delta = in->delta; f.x_zero_float = in->zero_float; f.x_eps = in->eps; f.x_n_bits = in->n_bits;
    }
    f.q_mid = *g.q_mid; f.w2 = g.w2_idx; f.rs2 = g.w2_rowsum; f.M = (uint32_t)M;
    LinArgs& a = f.lin2;
    a.w_rowsum = g.w2_rowsum; a.bias = g.bias2; a.y = last ? y : nullptr; a.y_idx = last ? y_idx : nullptr;
    a.M = (uint32_t)M; a.N = (uint32_t)N2; a.K = (uint32_t)N1;
    a.x_delta = g.q_mid->delta; a.x_zero_float = g.q_mid->zero_float; a.x_eps = g.q_mid->eps; a.x_n_bits = g.q_mid->n_bits;
    a.w_delta = g.w2_delta; a.w_n_params = (uint32_t)g.w2_n_params; a.w_eps = g.w2_eps; a.act = ACT_NONE;
    a.group_cols = (uint32_t)N2; a.fast_epi = fast_epi;
    a.tail = 2; a.residual = residual; a.nn_w = g.nn_weight; a.nn_b = g.nn_bias;
    a.has_q = g.q_dense != nullptr;
    if (g.q_dense) a.q_out = *g.q_dense;
    a.on_t1 = g.q_sum != nullptr;
    if (g.q_sum) a.q_t1 = *g.q_sum;
    a.on_t2 = g.q_out != nullptr;
    if (g.q_out) a.q_t2 = *g.q_out;
  }
  const int nw = tuning("TQ_FFN_CHAIN_WAVES", 8);
  TQ_REQUIRE(nw == 4 || nw == 8, "TQ_FFN_CHAIN_WAVES must be 4 or 8");
  // x tile | intermediate | W1 | W2 | column constants | index staging of the inner stages (16 rows x (N2 / nw + 16) B per wave)
  const size_t lds = (size_t)16 * 128 + 4 * (size_t)16 * 128 + 512 * 128 + 128 * 512 + 3 * 512 * 4 + 5 * 128 * 4 +
                     (size_t)nw * 16 * (128 / nw + 16);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)(M / 16));
  const int nf = (int)n_stages;
#ifdef TQ_FFN_PROF
  const char* pe = getenv("TQ_FFN_PROF_PTR");
  unsigned long long* prof = pe ? reinterpret_cast<unsigned long long*>(strtoull(pe, nullptr, 0)) : nullptr;
#define TQ_FPROF_PASS , prof
#else
#define TQ_FPROF_PASS
#endif
#define TQ_CHAIN(DT, W) hipLaunchKernelGGL((ffn_chain_i8_k<128, 512, 128, DT, W>), grid, dim3((W) * 64), lds, st, c[0], c[1], c[2], c[3], nf TQ_FPROF_PASS)
  if (y_dtype == TQ_F32) { if (nw == 8) TQ_CHAIN(TQ_F32, 8); else TQ_CHAIN(TQ_F32, 4); }
  else                   { if (nw == 8) TQ_CHAIN(TQ_BF16, 8); else TQ_CHAIN(TQ_BF16, 4); }
#undef TQ_CHAIN
#undef TQ_FPROF_PASS
  return check_launch("ffn_chain_i8_k");
}
