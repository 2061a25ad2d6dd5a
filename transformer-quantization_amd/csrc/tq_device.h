// Device-side helpers shared by the gfx950 kernels of libtq_hip.so.
// Wave = 64 lanes, 16-byte vector memory ops, fp32 register math (see include/tq_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/tq_hip.h"

namespace tq {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;
constexpr int kBlock = 256;          // 4 waves, one per SIMD
constexpr int kMaxGrid = 256 * 8;    // 256 CUs x 8 resident 256-thread blocks

// ------------------------------------------------------------------ quantizer parameters
// (scale, zero_point, int_min, int_max) exactly as the reference's properties compute them
// from the raw buffers: quantizers.py:132-153 (asym) and :321-332 (sym).
struct QP {
  float scale, zp, lo, hi;
};

__device__ __forceinline__ float grid_top(int n_bits) {
  // 2.0 ** n_bits - 1 evaluated in double like the python float, then narrowed to fp32
  return (float)(ldexp(1.0, n_bits) - 1.0);
}

// The same value from integer arithmetic (n_bits <= 24: the ABI's range) for the quantizer derivations of kernel prologues
// (round 6: three double-precision instructions per quantizer otherwise; a latency-bound fused launch derives up to six).
// NOT used for the explicit grid_top() calls of the integer Linear: with them in integer form the register allocator spilt
// the accumulators of its 128 x 128 tile kernel (272 bytes of scratch per lane, M = 8192: 45 -> 114 us).
__device__ __forceinline__ float grid_top_small(int n_bits) { return (float)((1u << (n_bits & 31)) - 1u); }

// scale = exp(delta) in the log domain, max(delta, eps) otherwise (quantizers.py:142-147).  log_domain is a wave-uniform
// kernel argument: a REAL branch -- written as a select, expf (~15 VALU instructions and up to 30 registers: the fused
// LayerNorm tails went from 3 to 4 waves per SIMD) is evaluated speculatively for every quantizer of every prologue.  The
// volatile asm keeps the compiler from if-converting the block back.  SELECT = true: the select form, for the 128 x 128
// tile integer Linear -- there the asm stops the unrolling of the epilogue loops and the accumulators end up in scratch
// memory (272 bytes per lane; M = 8192: 45 -> 114 us, caught by profiles/r06/kernel_table.md; guarded since by
// tests/test_abi.py::test_no_kernel_uses_scratch_memory).
template <bool SELECT = false>
__device__ __forceinline__ float effective_scale(int log_domain, float d, float eps) {
  if (SELECT) return log_domain ? expf(d) : (d < eps ? eps : d);
  float s = d < eps ? eps : d;
  if (log_domain) {
    s = expf(d);
    asm volatile("" : "+v"(s));
  }
  return s;
}

__device__ __forceinline__ float clamp_nanprop(float v, float lo, float hi) {
  // torch.clamp semantics: NaN stays NaN (both comparisons are false)
  v = v < lo ? lo : v;
  v = v > hi ? hi : v;
  return v;
}

template <bool SELECT = false>
__device__ __forceinline__ QP make_qp(const tq_quantizer& q, uint64_t p) {
  QP r;
  const float d = q.delta[p];
  r.scale = effective_scale<SELECT>(q.log_domain, d, q.eps);             // quantizers.py:142-147
  if (q.symmetric) {
    const bool sgn = q.signed_flag != nullptr && q.signed_flag[0] != 0;
    r.zp = 0.0f;                                                  // :330-332
    r.lo = sgn ? -ldexpf(1.0f, q.n_bits - 1) : 0.0f;         // :321-323
    r.hi = grid_top_small(q.n_bits - (sgn ? 1 : 0));                    // :325-328
  } else {
    r.lo = 0.0f;                                                  // :132-135
    r.hi = grid_top_small(q.n_bits);                                    // :137-140
    r.zp = clamp_nanprop(rintf(q.zero_float[p]), r.lo, r.hi);     // :149-153
  }
  return r;
}

// Touch every 64-byte line of the kernel-argument block once, all reads in flight together (ONE asm statement: the
// compiler does not track loads issued from inline asm, so the wait has to sit in the same statement as the loads).
// With large argument structs the compiler re-reads arguments from the block wherever they are needed and waits for
// each read; the first touch of a line is a scalar-cache miss (~300-500 cycles).  A latency-bound launch goes through
// a dozen of those one after the other -- measured in the chained feed-forward kernel: 2400-3100 cycles per stage for
// ~400 instructions of parameter set-up, 4.2 instead of 5.3 us per stage once the lines are resident.  BYTES: size of
// the explicit arguments (never read past it).
template <int BYTES>
__device__ __forceinline__ void prefetch_kernarg() {
  static_assert(BYTES > 0 && BYTES <= 4096, "explicit kernel arguments");
  const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
  uint32_t sink;
  asm volatile(".set tq_ka_off, 0\n\t.rept %2\n\ts_load_dword %0, %1, tq_ka_off\n\t.set tq_ka_off, tq_ka_off+64\n\t.endr\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&s"(sink)
               : "s"(ka), "n"((BYTES + 63) / 64 - (BYTES % 64 != 0 && BYTES % 64 < 4 ? 1 : 0))
               : "memory");
}

// The same in two halves, for kernels whose run time is their PROLOGUE (the fused launches of a model forward at
// inference batch sizes: a few microseconds in all).  make_qp reads delta, then -- behind the branches on the quantizer
// kind -- zero_float or the signed flag: two dependent memory round trips per quantizer, and a kernel with three
// quantizers that also derives its code path from them spent ~12 round trips (~3 us) before its first data load.
// load_qraw issues every read of one quantizer unconditionally (an absent buffer reads `safe` instead, any valid device
// address), so that the loads of ALL quantizers of a kernel -- and its first data loads -- are in flight together;
// qp_from_raw is make_qp's arithmetic on the values.  Same results as make_qp for every input.
struct QRaw {
  float d, z;
  uint32_t s;
};

__device__ __forceinline__ QRaw load_qraw(const tq_quantizer& q, uint64_t p, const float* safe) {
  const float* dp = q.delta != nullptr ? q.delta + p : safe;
  const float* zp = q.zero_float != nullptr ? q.zero_float + p : safe;
  const uint8_t* sp = q.signed_flag != nullptr ? q.signed_flag : reinterpret_cast<const uint8_t*>(safe);
  QRaw r;
  r.d = *dp;
  r.z = *zp;
  r.s = *sp;
  return r;
}

// Call on every QRaw of the kernel AFTER all of them (and the first data loads) were issued: pins the values to this point
// of the program.  Without it the compiler sinks each (single-use) load into the branch of qp_from_raw that consumes it,
// which brings back the dependent round trips one by one.
// (Wave-uniform parameters only: the values are moved to scalar registers.)
__device__ __forceinline__ float pinned_uniform(float v) {
  uint32_t b = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(uint32_t, v));
  asm volatile("" : "+s"(b));
  return __builtin_bit_cast(float, b);
}
__device__ __forceinline__ void qraw_arrived(QRaw& w) {
  w.d = pinned_uniform(w.d);
  w.z = pinned_uniform(w.z);
  uint32_t s = __builtin_amdgcn_readfirstlane(w.s);
  asm volatile("" : "+s"(s));
  w.s = s;
}

template <bool SELECT = false>
__device__ __forceinline__ QP qp_from_raw(const tq_quantizer& q, const QRaw& w) {
  QP r;
  r.scale = effective_scale<SELECT>(q.log_domain, w.d, q.eps);
  if (q.symmetric) {
    const bool sgn = q.signed_flag != nullptr && w.s != 0;
    r.zp = 0.0f;
    r.lo = sgn ? -ldexpf(1.0f, q.n_bits - 1) : 0.0f;
    r.hi = grid_top_small(q.n_bits - (sgn ? 1 : 0));
  } else {
    r.lo = 0.0f;
    r.hi = grid_top_small(q.n_bits);
    r.zp = clamp_nanprop(rintf(w.z), r.lo, r.hi);
  }
  return r;
}

// ------------------------------------------------------------------ rne(x / scale) without the division
// The IEEE fp32 division is 10 of the ~19 VALU instructions of a fake-quant element and makes the bf16
// per-embedding and the candidate-search kernels VALU-bound.  Only rne(x / scale) is needed, so:
//     r  = RN(1 / scale)                      once per parameter (true division)
//     q0 = RN(x * r),  h = rne(q0)            |q0 - RN(x / scale)| <= 3 u |x / scale|,  u = 2^-24
// h can differ from rne(RN(x / scale)) only if a rounding tie m + 1/2 lies within that distance of q0.
// The guard accepts h when q0 is farther than 4 u (|h| + 1) from every tie (q0 - h is exact); otherwise --
// a few 1e-5 of the elements of an 8-bit grid (the band grows with |h|), NaN / Inf, |h| >= 2^21, or a scale
// outside [2^-100, 2^100] (r = NaN) -- THAT element is recomputed with the true division.  The result is
// bit-identical to the division path for every input (tests/test_hip_parity.py: tie-adjacent inputs).
// Not used by the per-tensor kernels: they are HBM-bound with the division in place, and a data-dependent
// slow path can only cost there (measured: -20 % on a tensor whose indices span the whole 8-bit grid when
// a doubtful element made its whole 8-element vector take the division).
constexpr float kTieTol = 2.384185791015625e-07f;    // 4 u = 2^-22

__device__ __forceinline__ float guarded_rcp(float scale) {
  return (scale >= 7.888609052210118e-31f && scale <= 1.2676506002282294e30f) ? 1.0f / scale : __builtin_nanf("");
}

// rne(x / scale) given r = guarded_rcp(scale)
__device__ __forceinline__ float rne_quot1(float x, float scale, float r) {
  const float q0 = x * r;
  float h = rintf(q0);
  const float thr = __builtin_fmaf(fabsf(h), -kTieTol, 0.5f - kTieTol);
  if (!(fabsf(q0 - h) < thr)) h = rintf(x / scale);
  return h;
}

// V quotients sharing one scale
template <int V>
__device__ __forceinline__ void rne_quot(const float (&x)[V], float scale, float r, float (&h)[V]) {
#pragma unroll
  for (int j = 0; j < V; ++j) h[j] = rne_quot1(x[j], scale, r);
}

// ------------------------------------------------------------------ exact, branch-free, two elements per instruction
// gfx950 issues fp32 mul / add / fma on register PAIRS at full rate (v_pk_mul_f32, v_pk_add_f32, v_pk_fma_f32); rounding
// (v_rndne_f32) and median stay scalar.  Kernels that run several quantizers per element (fused layer tails) or many
// candidate quantizers per element (range search) are VALU-bound with the IEEE division (~11 issue slots) and were
// latency-bound with the guarded reciprocal above (a data-dependent branch per pair serialises the element chains).
// Two exact identities remove both, for every finite or infinite input:
//
//  (1) Correctly rounded quotient from a correctly rounded reciprocal (Markstein's theorem, fused multiply-add):
//          r = RN(1 / s)  (true division, once per parameter)
//          q0 = RN(x * r);   e = RN(x - q0 * s)  [fma, exact];   q1 = RN(q0 + e * r)  [fma]     ==>  q1 == RN(x / s)
//      3 packed instructions per pair instead of v_div_scale / v_rcp / 4 x v_fma / v_div_fmas / v_div_fixup per element.
//  (2) The clamp can be applied to x instead of to the index: with k_lo = lo - zp, k_hi = hi - zp and
//          y_lo = RN(s * k_lo), y_hi = RN(s * k_hi)            (the dequantised grid ends),
//      Q(x) = rne(RN(x / s)) is monotone and Q(y_hi) = k_hi, Q(y_lo) = k_lo (|k| < 2^22), hence
//          clamp(Q(x) + zp, lo, hi) - zp  ==  Q(med3(x, y_lo, y_hi)).
//      One v_med3_f32 bounds the operand (no overflow / Inf inside (1)) AND replaces the two-sided clamp; the
//      dequantised value is s * Q(.) because (h + zp) - zp == h exactly.
// => 4 issue slots per element: med3, pk_mul, 2 x pk_fma, rndne, pk_mul (pairs).  Both identities are checked with exact
// rational arithmetic on tie-adjacent and grid-end inputs in tests/test_exact_quotient.py, and through the kernels by
// tests/test_hip_parity.py::test_rounding_ties_are_bit_exact.  NaN is NOT preserved by v_med3 (it returns an operand that
// is not NaN): callers propagate NaN separately.  Valid for scales in [2^-100, 2^100] and grids below 2^22 steps
// (`ok`); kernels take their division path otherwise.
struct QF {
  f32x2 scale, nscale, rcp;
  float ylo, yhi, zp;
  bool ok;
};

__device__ __forceinline__ QF make_qf(const QP& p) {
  QF f;
  const float rc = guarded_rcp(p.scale);
  const float klo = p.lo - p.zp, khi = p.hi - p.zp;
  f.scale = f32x2{p.scale, p.scale};
  f.nscale = f32x2{-p.scale, -p.scale};
  f.rcp = f32x2{rc, rc};
  f.ylo = p.scale * klo;
  f.yhi = p.scale * khi;
  f.zp = p.zp;
  f.ok = (rc == rc) && fabsf(klo) < 4194304.0f && fabsf(khi) < 4194304.0f;
  return f;
}

// h = clamp(rne(x / s) + zp, lo, hi) - zp for two values (requires q.ok; NaN inputs give an unspecified grid value)
__device__ __forceinline__ f32x2 qf_round2(f32x2 x, const QF& q) {
  f32x2 xc;
  xc.x = __builtin_amdgcn_fmed3f(x.x, q.ylo, q.yhi);
  xc.y = __builtin_amdgcn_fmed3f(x.y, q.ylo, q.yhi);
  const f32x2 q0 = xc * q.rcp;
  const f32x2 e = __builtin_elementwise_fma(q0, q.nscale, xc);
  const f32x2 q1 = __builtin_elementwise_fma(e, q.rcp, q0);
  return f32x2{rintf(q1.x), rintf(q1.y)};
}
// fake-quantized pair: s * ((h + zp) - zp) == s * (h + 0): the addition turns h = -0 (x = -0) into the +0 the reference's
// `x_int - zero_point` produces, everything else is unchanged.  Evaluated as ONE fma, RN(s h + (+0)): identical to RN(s h)
// for every h != 0 (adding an exact zero changes nothing), and (-0) + (+0) = +0 under round-to-nearest for h = -0 --
// the same bits as the product of the normalised index, one packed instruction less per pair (round 4).
__device__ __forceinline__ f32x2 qf_dequant2(f32x2 h, const QF& q) {
  return __builtin_elementwise_fma(q.scale, h, f32x2{0.0f, 0.0f});
}
__device__ __forceinline__ f32x2 qf_fake_quant2(f32x2 x, const QF& q) {
  return qf_dequant2(qf_round2(x, q), q);
}

// N pairs, stage by stage: N independent dependency chains side by side in source order.  (A dependent packed op
// needs a wait state; written one pair after the other the scheduler kept the chains serial and emitted an s_nop after
// nearly every packed instruction.)
template <int N>
__device__ __forceinline__ void qf_round2_n(const f32x2 (&x)[N], const QF& q, f32x2 (&h)[N]) {
  f32x2 xc[N], q0[N], e[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    xc[i].x = __builtin_amdgcn_fmed3f(x[i].x, q.ylo, q.yhi);
    xc[i].y = __builtin_amdgcn_fmed3f(x[i].y, q.ylo, q.yhi);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) q0[i] = xc[i] * q.rcp;
#pragma unroll
  for (int i = 0; i < N; ++i) e[i] = __builtin_elementwise_fma(q0[i], q.nscale, xc[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) q0[i] = __builtin_elementwise_fma(e[i], q.rcp, q0[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) h[i] = f32x2{rintf(q0[i].x), rintf(q0[i].y)};
}
// The same without the "+ 0": for a quantizer whose output only feeds ANOTHER quantizer (through additions /
// multiplications), where a -0 instead of +0 cannot reach the final result (x + (-0) == x + 0 for x != 0, and a zero
// that survives to the last quantizer is normalised there).  Saves one packed add per pair in the fused layer tails.
template <int N>
__device__ __forceinline__ void qf_fake_quant2_n_signed_zero(f32x2 (&x)[N], const QF& q) {
  f32x2 h[N];
  qf_round2_n<N>(x, q, h);
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = q.scale * h[i];
}
template <int N>
__device__ __forceinline__ void qf_fake_quant2_n(f32x2 (&x)[N], const QF& q) {
  f32x2 h[N];
  qf_round2_n<N>(x, q, h);
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = qf_dequant2(h[i], q);                      // -0 -> +0: see qf_fake_quant2
}

// x_int = clamp(round(x / scale) + zp, lo, hi)   (quantizers.py:184-185)
__device__ __forceinline__ float q_index(float x, const QP& p) {
  return clamp_nanprop(rintf(x / p.scale) + p.zp, p.lo, p.hi);
}
// y = scale * (x_int - zp)                        (quantizers.py:209)
__device__ __forceinline__ float q_dequant(float xi, const QP& p) {
  return p.scale * (xi - p.zp);
}

// ------------------------------------------------------------------ exp(x), x <= 0, cheap and accurate
// For the fused softmax (csrc/tq_fused_ln.hip), whose contract is a TOLERANCE (tests/test_fused_softmax.py: 1e-5 relative,
// >= 99.9 % identical indices behind the probability quantizer) -- not the bit-exact `exp_neg_ieee` of the integer
// attention core below.  libm's expf costs ~13 issue slots (range reduction, ldexp, two range checks with selects); here
// the hardware 2^y does the range reduction and the rounding error of y = x log2(e) is repaired to first order:
//   yh = RN(x c_hi), yl = (x c_hi - yh) [exact, one fma] + x c_lo;   e^x = 2^yh (1 + yl ln 2 + O(yl^2)),  |yl| < 2^-17
// 8 slots, < 2 ulp for results >= 2^-126; smaller results (and x = -inf, where yl is inf - inf) return 0.  NaN -> NaN.
__device__ __forceinline__ float exp_nonpos_fast(float x) {
  const float yh = x * 1.44269502162933349609375f;
  float yl = __builtin_fmaf(x, 1.44269502162933349609375f, -yh);
  yl = __builtin_fmaf(x, 1.92596299112661746e-8f, yl);
  const float e = __builtin_amdgcn_exp2f(yh);
  const float r = __builtin_fmaf(e * 0.693147182464599609375f, yl, e);
  return e == 0.0f ? 0.0f : r;
}

// max(a, b) as ONE v_max_f32: fmaxf() under IEEE semantics is preceded by a canonicalising `v_max_f32 v, v, v` per operand
// (quiets signalling NaNs), which doubles the instruction count of a max reduction; quiet-NaN behaviour is the same
// (the non-NaN operand wins).
__device__ __forceinline__ float max_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// ------------------------------------------------------------------ exp(x), x <= 0, from IEEE operations only
// The softmax of the integer attention core is specified by THIS function (and its plain-C twin tq_exp_neg in
// oracle/tq_int_oracle.c) rather than by libm's expf, whose v_exp_f32 core no CPU reproduces: Cephes-style range
// reduction x = k ln2 + r (two fma with a split ln2), degree-5 polynomial for e^r - 1 - r, exact scaling by 2^k.
// mul / rndne / fma / add / ldexp are all correctly rounded, so kernel and oracle agree bit for bit and the whole
// attention core (exact integer contractions around it) is testable at zero tolerance.  x < -86 (incl. -inf) -> 0:
// keeps 2^k y in the normal range; NaN -> NaN.  < 2 ulp on [-86, 0]; ~16 issue slots (libm's expf: ~14).
__device__ __forceinline__ float exp_neg_ieee(float x) {
  const float k = rintf(x * 1.44269504088896341f);
  float r = __builtin_fmaf(k, -0.693359375f, x);
  r = __builtin_fmaf(k, 2.12194440e-4f, r);
  const float z = r * r;
  float y = __builtin_fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
  y = __builtin_fmaf(y, r, 8.3334519073e-3f);
  y = __builtin_fmaf(y, r, 4.1665795894e-2f);
  y = __builtin_fmaf(y, r, 1.6666665459e-1f);
  y = __builtin_fmaf(y, r, 5.0000001201e-1f);
  y = __builtin_fmaf(y, z, r);
  y = y + 1.0f;
  y = ldexpf(y, (int)k);
  // NaN needs no select of its own: k, r and y are NaN with x (rndne / fma propagate it), v_cvt_i32_f32(NaN) = 0, so
  // ldexp(NaN, 0) = NaN, and `x < -86` is false for NaN -- the oracle's explicit `x != x ? x : y` returns the same bits' class
  return x < -86.0f ? 0.0f : y;
}

// ------------------------------------------------------------------ storage <-> fp32
// NB: always bit-cast a by-value scalar.  clang (ROCm 7.2) mis-compiles
// __builtin_bit_cast(T, vec[i]) on an ext_vector element lvalue: every i reads element 0.
__device__ __forceinline__ float bits_to_f32(uint32_t w) { return __builtin_bit_cast(float, w); }
__device__ __forceinline__ uint32_t f32_to_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

// ------------------------------------------------------------------ activation + quantizer as a staircase
// h(v) = Q(act(v)) - zero_point of a <= 8-bit quantizer is a step function of the fp32 pre-activation v with at most 255
// steps.  csrc/tq_stair.hip tabulates it over uniform bins of v that hold at most ONE step each; a consumer (the integer
// Linear's epilogue) then needs one fma + clamp + convert for the bin, one 8-byte table read, one compare and a select
// instead of evaluating the activation and the quantizer's quotient (with GELU: ~69 instead of ~100 issue cycles per
// output-instruction, profiles/r04/valu_probe2.txt; 26.7 -> 21.6 VALU instructions per output over the whole kernel).
// Table: [StairHdr | nb entries {T, packed}], packed = bf16(h right of T) << 16 | bf16(h left of T) (|h| <= 256
// is exact in bf16), h(v) = v >= T ? right : left inside the bin.  `stair_bin` is THE bin map: the builder derives
// every bin's fp32 interval from this very function, so builder and consumers agree on every input bit pattern.
struct StairHdr {
  float inv_w, c0, nbm1;   // bin = trunc(clamp(fma(v, inv_w, c0), 0, nbm1))
  float ok;                // 1 = the table is exact for every finite fp32 v; 0 = use the arithmetic path
};
__device__ __forceinline__ uint32_t stair_bin(float v, float inv_w, float c0, float nbm1) {
  return (uint32_t)__builtin_amdgcn_fmed3f(__builtin_fmaf(v, inv_w, c0), 0.0f, nbm1);   // NaN -> bin 0
}
__device__ __forceinline__ float stair_pick(float v, float T, uint32_t packed) {
  const uint32_t w = v >= T ? packed : (packed << 16);
  return bits_to_f32(w & 0xffff0000u);
}

template <int DT> struct Store;   // DT = TQ_F32 / TQ_BF16 / TQ_F16

template <> struct Store<TQ_F32> {
  typedef float elem_t;
  static constexpr int kVec = 4;   // elements per 16-byte vector
  static __device__ __forceinline__ void unpack(const u32x4& v, float (&f)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { const uint32_t w = v[i]; f[i] = bits_to_f32(w); }
  }
  static __device__ __forceinline__ u32x4 pack(const float (&f)[4]) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = f32_to_bits(f[i]);
    return v;
  }
  static __device__ __forceinline__ float load1(const elem_t* p) { return *p; }
  static __device__ __forceinline__ void store1(elem_t* p, float f) { *p = f; }
};

template <> struct Store<TQ_BF16> {
  typedef uint16_t elem_t;
  static constexpr int kVec = 8;
  static __device__ __forceinline__ void unpack(const u32x4& v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t w = v[i];
      f[2 * i] = bits_to_f32(w << 16);
      f[2 * i + 1] = bits_to_f32(w & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ u32x4 pack(const float (&f)[8]) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x2 t = {f[2 * i], f[2 * i + 1]};
      v[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(t, bf16x2));  // v_cvt_pk_bf16_f32 (RNE)
    }
    return v;
  }
  static __device__ __forceinline__ float load1(const elem_t* p) {
    return __builtin_bit_cast(float, (uint32_t)(*p) << 16);
  }
  static __device__ __forceinline__ void store1(elem_t* p, float f) {
    *p = __builtin_bit_cast(uint16_t, (__bf16)f);
  }
};

// fp32 -> fp16 narrowing must see the ROUNDED fp32 value: without the barrier LLVM folds the producing multiply
// into v_fma_mixlo_f16, which rounds the exact product once and differs from RNE(fp32 result) at double-rounding
// ties (found by tests/test_fuzz_parity.py: 46 of 442k elements of a per-embedding fp16 case).
__device__ __forceinline__ _Float16 narrow_f16(float f) {
  asm volatile("" : "+v"(f));
  return (_Float16)f;
}

template <> struct Store<TQ_F16> {
  typedef uint16_t elem_t;
  static constexpr int kVec = 8;
  static __device__ __forceinline__ void unpack(const u32x4& v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t w = v[i];
      const f16x2 h = __builtin_bit_cast(f16x2, w);
      f[2 * i] = (float)h[0];
      f[2 * i + 1] = (float)h[1];
    }
  }
  static __device__ __forceinline__ u32x4 pack(const float (&f)[8]) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f16x2 h = {narrow_f16(f[2 * i]), narrow_f16(f[2 * i + 1])};
      v[i] = __builtin_bit_cast(uint32_t, h);
    }
    return v;
  }
  static __device__ __forceinline__ float load1(const elem_t* p) {
    return (float)__builtin_bit_cast(_Float16, *p);
  }
  static __device__ __forceinline__ void store1(elem_t* p, float f) {
    *p = __builtin_bit_cast(uint16_t, narrow_f16(f));
  }
};

// streaming (read-once / write-once) 16-byte accesses
__device__ __forceinline__ u32x4 ld_stream(const u32x4* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(u32x4* p, u32x4 v) { __builtin_nontemporal_store(v, p); }

// ------------------------------------------------------------------ wave / block reductions
// torch.min / torch.max propagate NaN; so do these.
__device__ __forceinline__ float min_nanprop(float a, float b) {
  const float m = b < a ? b : a;   // a NaN -> compare false -> m = a (NaN)
  return (b != b) ? b : m;
}
__device__ __forceinline__ float max_nanprop(float a, float b) {
  const float m = b > a ? b : a;
  return (b != b) ? b : m;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min_nanprop(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max_nanprop(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}


// ------------------------------------------------------------------ running min / max
constexpr float kInf = __builtin_huge_valf();

// Branch-free running min/max with torch's NaN propagation: the compares ignore NaN, a third
// lane-local word remembers the largest |bits| seen (> 0x7f800000 <=> some input was NaN).
struct MinMax {
  float mn = kInf, mx = -kInf;
  uint32_t top = 0;
  __device__ __forceinline__ void add(float x) {
    mn = x < mn ? x : mn;
    mx = x > mx ? x : mx;
    const uint32_t a = f32_to_bits(x) & 0x7fffffffu;
    top = a > top ? a : top;
  }
  __device__ __forceinline__ float lo() const { return top > 0x7f800000u ? __builtin_nanf("") : mn; }
  __device__ __forceinline__ float hi() const { return top > 0x7f800000u ? __builtin_nanf("") : mx; }
};

}  // namespace tq
