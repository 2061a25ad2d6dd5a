// (f3) Fixed-range quantized self-attention core on the i8 matrix cores.
//
// Reference chain (models/quantized_bert.py:135-213), all on dequantised fp32 tensors with one ATen
// launch per step:   Q, K, V = quantized Linear outputs, split into heads (3 permute copies)
//     S = Q K^T                      batched fp32 GEMM            [B, H, T, T]
//     S = Q_scores(S) / sqrt(d) + mask ;  P = Q_probs(softmax(S))  5 element-wise sweeps
//     C = P V                        batched fp32 GEMM, permute copy back to [B, T, H d]
//     C = Q_ctx(C)
// With fixed per-tensor asymmetric ranges Q, K, V and P live on <= 8-bit grids, so both GEMMs are exact
// integer contractions of the grid indices (a - z) that the producing kernels already emit as
// int8(index - 128):
//     S[q,k] = s_q s_k ( sum_d a'_q a'_k + c_k sum_d a'_q + c_q sum_d a'_k + d c_q c_k ),   c = 128 - z
//     C[q,d] = s_p s_v ( sum_k a'_p a'_v + c_v sum_k a'_p + c_p sum_k a'_v + T c_p c_v )
// One workgroup (2 waves, 8 on grids that are not key-split: round 6) owns 32 (128) query rows of one (batch, head); each
// wave computes S^T = K Q^T for
// its 16 queries with T/16 v_mfma_i32_16x16x64_i8 (d = 64 = one MFMA K step), keeps the 16 x T scores
// in registers (lane = one query column, keys 16t + 4g + r), does the quantizers / mask / softmax
// there, packs the probability indices straight into the B operand of the second MFMA (the MFMA K
// dimension may be permuted freely as long as both operands agree, so the accumulator layout IS the
// operand layout once V^T is stored with the matching key permutation in LDS) and computes
// C^T = V^T P^T.  Row / column sums for the zero-point corrections are MFMAs against an all-ones
// operand.  Nothing but Q, K, V indices (3 x 64 B per token and head) is read and only C is written:
// the [B, H, T, T] score and probability tensors never exist in memory.
#include <algorithm>
#include <type_traits>

#include "tq_device.h"
#include "tq_host.h"

namespace tq {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kAttnWaves = 2;
constexpr int kAttnThreads = kAttnWaves * kWave;

struct AttnArgs {
  const int8_t *q, *k, *v;    // [B, T, H * 64] int8(index - 128)
  float* ctx;                 // [B, T, H * 64]
  int8_t* ctx_idx;            // optional int8(index - 128) of ctx (needs q_ctx)
  const float* mask;          // additive [B, T] or null
  uint32_t B, T, H;
  uint32_t in_stride;         // elements between consecutive tokens of q / k (H * 64, or 3 * H * 64 inside a stacked QKV buffer)
  uint32_t v_stride;          // the same for v (MobileBERT: Q | K stacked, V on its own)
  float denom;
  tq_quantizer qq, qk, qv;    // per-tensor asymmetric, n_bits <= 8
  tq_quantizer q_scores, q_probs, q_ctx;
  int has_scores, has_ctx;
#ifdef TQ_ATTN_PROF
  unsigned long long* prof;   // tools/tuning/attn_prof.py: 8 s_memtime stamps per workgroup
#endif
  int fast_ok;                // 0: keep the guarded-reciprocal element math (TQ_ATTN_FAST=0, parity tests)
};

// position of key (64 s + 16 tt + 4 g + r) inside a V^T row: 64 s + 16 g + 4 tt + r
__device__ __forceinline__ uint32_t key_slot(uint32_t key) {
  return (key & ~63u) | (((key >> 2) & 3u) << 4) | (((key >> 4) & 3u) << 2) | (key & 3u);
}

// RN(x / b) for N pairs given r = RN(1 / b), nb = -b (Markstein: exact fma residual, see tq_device.h); chunks of 8 pairs
// keep the live stage arrays small
template <int N>
__device__ __forceinline__ void quot2_n(f32x2* x, f32x2 r, f32x2 nb) {
  constexpr int C = N < 4 ? N : 4;
  static_assert(N % C == 0, "pairs");
#pragma unroll
  for (int c = 0; c < N; c += C) {
    f32x2 q0[C], e[C];
#pragma unroll
    for (int i = 0; i < C; ++i) q0[i] = x[c + i] * r;
#pragma unroll
    for (int i = 0; i < C; ++i) e[i] = __builtin_elementwise_fma(q0[i], nb, x[c + i]);
#pragma unroll
    for (int i = 0; i < C; ++i) x[c + i] = __builtin_elementwise_fma(e[i], r, q0[i]);
  }
}

// exp_neg_ieee (tq_device.h) for N register pairs: the same single IEEE operations per element -- v_pk_mul / v_pk_fma /
// v_pk_add ARE the scalar operations on two lanes -- so every result is bit-identical to the scalar function and to
// tq_exp_neg of oracle/tq_int_oracle.c; 10.5 instead of 16 issue slots per element (round 6: the exponential was 30 % of the
// VALU instructions of a launch, profiles/r06/attention_pmc_*.json).  Stage by stage over chunks of 8 pairs.
template <int N>
__device__ __forceinline__ void exp_neg_ieee2_n(f32x2* x) {
  constexpr int C = N < 4 ? N : 4;
  static_assert(N % C == 0, "pairs");
  auto k2 = [](float c) { return f32x2{c, c}; };
#pragma unroll
  for (int c = 0; c < N; c += C) {
    f32x2 k[C], r[C], y[C];
#pragma unroll
    for (int i = 0; i < C; ++i) k[i] = x[c + i] * k2(1.44269504088896341f);
#pragma unroll
    for (int i = 0; i < C; ++i) k[i] = f32x2{rintf(k[i].x), rintf(k[i].y)};
#pragma unroll
    for (int i = 0; i < C; ++i) r[i] = __builtin_elementwise_fma(k[i], k2(-0.693359375f), x[c + i]);
#pragma unroll
    for (int i = 0; i < C; ++i) r[i] = __builtin_elementwise_fma(k[i], k2(2.12194440e-4f), r[i]);
#pragma unroll
    for (int i = 0; i < C; ++i) y[i] = __builtin_elementwise_fma(k2(1.9875691500e-4f), r[i], k2(1.3981999507e-3f));
#pragma unroll
    for (int i = 0; i < C; ++i) y[i] = __builtin_elementwise_fma(y[i], r[i], k2(8.3334519073e-3f));
#pragma unroll
    for (int i = 0; i < C; ++i) y[i] = __builtin_elementwise_fma(y[i], r[i], k2(4.1665795894e-2f));
#pragma unroll
    for (int i = 0; i < C; ++i) y[i] = __builtin_elementwise_fma(y[i], r[i], k2(1.6666665459e-1f));
#pragma unroll
    for (int i = 0; i < C; ++i) y[i] = __builtin_elementwise_fma(y[i], r[i], k2(5.0000001201e-1f));
#pragma unroll
    for (int i = 0; i < C; ++i) y[i] = __builtin_elementwise_fma(y[i], r[i] * r[i], r[i]);
#pragma unroll
    for (int i = 0; i < C; ++i) y[i] = y[i] + k2(1.0f);
#pragma unroll
    for (int i = 0; i < C; ++i) {
      const float ex = ldexpf(y[i].x, (int)k[i].x), ey = ldexpf(y[i].y, (int)k[i].y);
      x[c + i].x = x[c + i].x < -86.0f ? 0.0f : ex;       // (NaN: see exp_neg_ieee)
      x[c + i].y = x[c + i].y < -86.0f ? 0.0f : ey;
    }
  }
}

// 4 x 4 byte transpose of four dwords: o[e] = bytes e of (r0, r1, r2, r3), r0 in the low byte.  v_perm_b32 selects
// bytes 0..3 from its SECOND source and 4..7 from its first: 8 instructions (shifts / masks / ors: ~24).
__device__ __forceinline__ void transpose4x4_b8(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t (&o)[4]) {
  const uint32_t t0 = __builtin_amdgcn_perm(r1, r0, 0x05010400u);      // r0.b0 r1.b0 r0.b1 r1.b1
  const uint32_t t1 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);      // r0.b2 r1.b2 r0.b3 r1.b3
  const uint32_t t2 = __builtin_amdgcn_perm(r3, r2, 0x05010400u);
  const uint32_t t3 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
  o[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
  o[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
  o[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
  o[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}

#ifdef TQ_ATTN_PROF
#define TQ_STAMP(k)                                                                                   \
  do {                                                                                                \
    unsigned long long t_;                                                                            \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    if (threadIdx.x == 0 && p.prof) p.prof[blockIdx.x * 8 + (k)] = t_;                                \
  } while (0)
#else
#define TQ_STAMP(k)
#endif

// value of the other key-half wave (same queries) through LDS; one barrier
__device__ __forceinline__ float pair_exchange(float (*slot)[16], float v, int wave, int kh, int r16, int g) {
  if (g == 0) slot[kh * 2 + wave][r16] = v;
  __syncthreads();
  return slot[(kh ^ 1) * 2 + wave][r16];
}

// SPLIT: four waves per 32 queries -- wave (qh, kh) owns 16 queries x one half of the keys.  One wave issues one VALU
// instruction per 4 cycles, and a 16 x T score tile is ~45 T instructions per lane: with B * H * T / 32 workgroups below
// ~2 waves per SIMD (BERT-base at batch 8: 0.75) the kernel is the latency of that one chain, so halving it pays for
// the three exchanges through LDS (row max, row sum, the integer partial sums of the second GEMM -- all exact, the
// float sum a + b is the same value in both waves).
// (Round 6, measured and not kept -- profiles/r06/attn_qt_ab.txt: a wave taking TWO 16-query tiles one after the other, V^T
// staging and parameter derivation paid once per 32 queries: 1561 -> 1415 instructions per tile but 165 registers, 3 waves
// per SIMD instead of 4: -4 % at B = 64, +11 % at B = 128 and at T = 256.)
// QW = query waves per workgroup of the one-wave-per-row form (2, or 8 on large grids: round 6).  With 8 waves a workgroup
// owns 128 queries of its (batch, head): the V tile is fetched and transposed into LDS once per 128 queries instead of
// once per 32, and the K tiles of the eight waves come from the same CU's L1.  Same per-wave code and data: bit-identical.
template <int NT_ALL, int DH, bool SPLIT, int QW = kAttnWaves>   // NT_ALL = T / 16 key tiles, DH = head dim (32 or 64)
__global__ __launch_bounds__(SPLIT ? 2 * kAttnThreads : QW * kWave) void attention_i8_k(AttnArgs p) {
  static_assert(QW == kAttnWaves || !SPLIT, "the key-split form has two query waves");
  static_assert((NT_ALL * 16) % (16 * QW) == 0, "whole workgroups per row of queries");
  constexpr int T = NT_ALL * 16;
  constexpr int NT = SPLIT ? NT_ALL / 2 : NT_ALL;  // key tiles of this wave
  constexpr int KS = NT / 4;                       // its 64-key MFMA steps of the second GEMM
  constexpr int THREADS = SPLIT ? 2 * kAttnThreads : QW * kWave;
  constexpr int PITCH = T + 32;                    // 32 * odd bytes: conflict-free ds_read_b128 (4 x 16 lane groups, 64 banks)
  constexpr int KSA = SPLIT ? 2 * KS : KS;         // 64-key steps of the wave that finishes the tile (all keys)
  static_assert(!SPLIT || NT_ALL % 8 == 0, "key split needs an even number of 64-key steps");
  __shared__ __attribute__((aligned(16))) int8_t s_vt[DH * PITCH];
  __shared__ float s_red[SPLIT ? 2 : 1][4][16];    // [max | sum][wave][query]
  // key-split form: the kh = 1 wave hands its probability indices (KS operands of 16 bytes per lane) to its kh = 0 partner
  __shared__ __attribute__((aligned(16))) v4i s_fp[SPLIT ? 2 : 1][SPLIT ? KS : 1][64];
  // QW = 8: the K tile of the (batch, head) is fetched ONCE per workgroup (one 16-byte load per thread at T = 128) and read
  // from LDS by the eight waves -- each wave fetching its own copy was 8 of the 13 loads per lane in front of the first
  // instruction, on a launch whose longest phase is that fetch (profiles/r06/attn_phase_profile.txt).  Row pitch DH + 16:
  // conflict-free ds_read_b128 of 16 rows x 16 bytes.
  constexpr bool K_LDS = !SPLIT && QW == 8;
  constexpr int KP = DH + 16;
  __shared__ __attribute__((aligned(16))) int8_t s_k[K_LDS ? T * KP : 16];

  const int tid = threadIdx.x, lane = tid & 63, wave = SPLIT ? (tid >> 6) & 1 : tid >> 6, kh = SPLIT ? tid >> 7 : 0;
  const int r16 = lane & 15, g = lane >> 4;
  const int t0 = kh * NT;                          // first key tile of this wave
  const uint32_t qblocks = T / (16 * QW);
  const uint32_t bh = blockIdx.x / qblocks, qb = blockIdx.x % qblocks;
  const uint32_t b = bh / p.H, h = bh % p.H;
  const size_t row_stride = p.in_stride;
  const size_t base = (size_t)b * T * row_stride + (size_t)h * DH;
  const size_t v_stride = p.v_stride, base_v = (size_t)b * T * v_stride + (size_t)h * DH;

  TQ_STAMP(0);
  // ---- V tile: loads first (one work item = 4 consecutive keys x 16 head dims, four 16-byte loads) ----------------
  constexpr uint32_t PARTS = DH / 16, ITEMS = (uint32_t)T / 4 * PARTS;
  constexpr int VIT = (ITEMS + THREADS - 1) / THREADS;
  v4i raw[VIT][4];
#pragma unroll
  for (int it = 0; it < VIT; ++it) {
    const uint32_t c = tid + it * THREADS;
    if (ITEMS % THREADS == 0 || c < ITEMS) {
      const uint32_t key4 = (c / PARTS) * 4, part = c % PARTS;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        raw[it][kk] = *reinterpret_cast<const v4i*>(p.v + base_v + (size_t)(key4 + kk) * v_stride + part * 16);
    }
  }

  // ---- quantizer parameters and this wave's queries: two dependent rounds of loads (kernel argument -> pointer ->
  // value) in flight together with the V tile
  const v4i ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
  const v4i zero4 = {0, 0, 0, 0};
  const uint32_t qrow = qb * 16 * QW + wave * 16 + r16;
  const bool kin = g * 16 < DH;                     // lane groups beyond the head dim supply zeros
  v4i fq = zero4;
  if (kin) fq = *reinterpret_cast<const v4i*>(p.q + base + (size_t)qrow * row_stride + g * 16);
  // short rows: the K tiles too -- one exposed memory latency for V, Q, K and the parameters instead of two
  constexpr bool K_EARLY = NT <= 8 && !K_LDS;
  v4i fk_all[K_EARLY ? NT : 1];
  constexpr uint32_t KCH = (uint32_t)T * PARTS;     // 16-byte chunks of the K tile (K_LDS)
  constexpr int KIT = K_LDS ? (KCH + THREADS - 1) / THREADS : 1;
  v4i kraw[KIT];
  if (K_LDS) {
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const uint32_t c = tid + it * THREADS;
      if (KCH % THREADS == 0 || c < KCH)
        kraw[it] = *reinterpret_cast<const v4i*>(p.k + base + (size_t)(c / PARTS) * row_stride + (c % PARTS) * 16);
    }
  }
  if (K_EARLY) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      fk_all[t] = zero4;
      if (kin) fk_all[t] = *reinterpret_cast<const v4i*>(p.k + base + (size_t)((t0 + t) * 16 + r16) * row_stride + g * 16);
    }
  }

  // the raw buffers of all six quantizers as ONE batch of independent loads (tq_device.h load_qraw; quantizer by
  // quantizer, delta -> zero_float, this was ~10 dependent scalar round trips: 3-4 us of a 10 us launch at BERT-base's
  // inference batch), pinned behind the V / Q / K loads
  QRaw wq = load_qraw(p.qq, 0, p.qq.delta), wk = load_qraw(p.qk, 0, p.qq.delta), wv = load_qraw(p.qv, 0, p.qq.delta);
  QRaw wp = load_qraw(p.q_probs, 0, p.qq.delta), ws = load_qraw(p.q_scores, 0, p.qq.delta), wc = load_qraw(p.q_ctx, 0, p.qq.delta);
  qraw_arrived(wq); qraw_arrived(wk); qraw_arrived(wv); qraw_arrived(wp); qraw_arrived(ws); qraw_arrived(wc);
  const QP pq = qp_from_raw(p.qq, wq), pk = qp_from_raw(p.qk, wk), pv = qp_from_raw(p.qv, wv), pp = qp_from_raw(p.q_probs, wp);
  const int cq = 128 - (int)pq.zp, ck = 128 - (int)pk.zp, cv = 128 - (int)pv.zp, cp = 128 - (int)pp.zp;
  const float s_qk = pq.scale * pk.scale, s_pv = pp.scale * pv.scale;
  QP ps = {1.f, 0.f, 0.f, 0.f}, pc = {1.f, 0.f, 0.f, 0.f};
  if (p.has_scores) ps = qp_from_raw(p.q_scores, ws);
  if (p.has_ctx) pc = qp_from_raw(p.q_ctx, wc);
  const float rcp_s = guarded_rcp(ps.scale), rcp_p = guarded_rcp(pp.scale);   // rne(x / scale), tq_device.h

  // denom = 2^k (normal range): the division is an exact scaling
  const uint32_t dbits = f32_to_bits(p.denom);
  const bool denom_pow2 = (dbits & 0x007fffffu) == 0 && (dbits >> 23) >= 32 && (dbits >> 23) <= 222;
  const float inv_denom = 1.0f / p.denom;

  // ---- V^T -> LDS with the key permutation of the accumulator layout: 4x4 byte transposes in registers, sixteen
  // 32-bit LDS stores per work item (keys 4m .. 4m+3 are adjacent slots of one V^T row)
#pragma unroll
  for (int it = 0; it < VIT; ++it) {
    const uint32_t c = tid + it * THREADS;
    if (ITEMS % THREADS == 0 || c < ITEMS) {
      const uint32_t key4 = (c / PARTS) * 4, part = c % PARTS;
      const uint32_t slot = key_slot(key4);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        uint32_t word[4];                              // word[e]: byte kk = head dim 4 w + e of key key4 + kk
        transpose4x4_b8((uint32_t)raw[it][0][w], (uint32_t)raw[it][1][w], (uint32_t)raw[it][2][w], (uint32_t)raw[it][3][w], word);
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<uint32_t*>(s_vt + (part * 16 + w * 4 + e) * PITCH + slot) = word[e];
      }
    }
  }

  if (K_LDS) {
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const uint32_t c = tid + it * THREADS;
      if (KCH % THREADS == 0 || c < KCH) *reinterpret_cast<v4i*>(s_k + (c / PARTS) * KP + (c % PARTS) * 16) = kraw[it];
    }
    __syncthreads();                                 // K tile and V^T are in LDS
  }
  TQ_STAMP(1);
  // ---- S^T = K Q^T for this wave's 16 queries --------------------------------------------------------
  const int rsq = __builtin_amdgcn_mfma_i32_16x16x64_i8(ones, fq, zero4, 0, 0, 0)[0];   // sum_d a'_q of column r16
  const int q_const = ck * rsq + DH * cq * ck;

  // Wave-uniform: every quantizer of the score / probability chain admits the exact branch-free path (tq_device.h QF)
  // and the Markstein quotients below keep their fma residuals exact (scales within 2^+-60, see tq_fused_ln.hip).
  const QF fs = make_qf(ps), fpq = make_qf(pp);
  const float adn = fabsf(p.denom);
  const bool fast = p.fast_ok && fpq.ok && pp.scale >= 0x1p-60f && pp.scale <= 0x1p60f && adn >= 0x1p-20f && adn <= 0x1p20f &&
                    s_qk >= 0x1p-60f && s_qk <= 0x1p60f &&
                    (!p.has_scores || (fs.ok && ps.scale >= 0x1p-60f && ps.scale <= 0x1p60f));

  TQ_STAMP(2);
  const QF fc = make_qf(pc);
  const bool fast_ctx = p.fast_ok && p.has_ctx && fc.ok;

  // The zero-point corrections ride on the matrix cores (round 6: they were a v_mul_lo + v_add3 per score): the per-query
  // constant is the accumulator's initial value, and c_q sum_d a'_k is one more MFMA of the K tile against an operand whose
  // bytes are all c_q (c_q = 128 - z_q is in [-127, 128]; 128 = two passes with 64).  Same exact integers as before.
  const int cq_b = cq == 128 ? 64 : cq;
  const int cq_w = (int)((uint32_t)(cq_b & 0xff) * 0x01010101u);
  const v4i cq4 = {cq_w, cq_w, cq_w, cq_w};
  const v4i qc4 = {q_const, q_const, q_const, q_const};
  float sc[NT][4];
  auto score_tiles = [&](auto wide) {                // (one wave-uniform branch around the loop, not one per tile)
    if constexpr (K_EARLY) {
      // register-resident K tiles: the NT accumulator chains side by side, pass by pass (a dependent MFMA waits for
      // its predecessor's last pass: tile by tile the second MFMA of every tile stalled the wave)
      v4i acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fk_all[t], fq, qc4, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fk_all[t], cq4, acc[t], 0, 0, 0);
      if (decltype(wide)::value) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fk_all[t], cq4, acc[t], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[t][r] = (float)acc[t][r] * s_qk;
      return;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      v4i fk = zero4;
      if (K_EARLY) fk = fk_all[t];
      else if (K_LDS) { if (kin) fk = *reinterpret_cast<const v4i*>(s_k + ((t0 + t) * 16 + r16) * KP + g * 16); }
      else if (kin) fk = *reinterpret_cast<const v4i*>(p.k + base + (size_t)((t0 + t) * 16 + r16) * row_stride + g * 16);
      v4i acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(fk, fq, qc4, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(fk, cq4, acc, 0, 0, 0);               // + c_q sum_d a'_k of rows 4g + r
      if (decltype(wide)::value) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(fk, cq4, acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[t][r] = (float)acc[r] * s_qk;
      // pin the four scores: the scheduler otherwise keeps the integer accumulators of every tile alive
      asm volatile("" : "+v"(sc[t][0]), "+v"(sc[t][1]), "+v"(sc[t][2]), "+v"(sc[t][3]));
    }
  };
  if (cq == 128) score_tiles(std::true_type{});
  else score_tiles(std::false_type{});

  TQ_STAMP(3);
  v4i fp[KS];
  bool bad_row;                                      // NaN row sum (fully masked query): the context row is NaN
  if (NT <= 16 && fast) {       // (longer rows: the stage arrays would spill)
    // ---- branch-free: NT * 2 register pairs move through every stage side by side ---------------------------------
    constexpr int P = NT * 2;
    f32x2 x[P];
#pragma unroll
    for (int t = 0; t < NT; ++t) { x[2 * t] = f32x2{sc[t][0], sc[t][1]}; x[2 * t + 1] = f32x2{sc[t][2], sc[t][3]}; }
    if (p.has_scores) {
#pragma unroll
      for (int c = 0; c < P; c += 4) {
        f32x2 (&xc)[4] = *reinterpret_cast<f32x2(*)[4]>(&x[c]);
        qf_fake_quant2_n<4>(xc, fs);
      }
    }
    if (denom_pow2) {
      const f32x2 rd = {inv_denom, inv_denom};
#pragma unroll
      for (int i = 0; i < P; ++i) x[i] = x[i] * rd;
    } else {                                         // RN(x / denom)
      quot2_n<P>(x, f32x2{inv_denom, inv_denom}, f32x2{-p.denom, -p.denom});
    }
    if (p.mask) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + (size_t)b * T + (t0 + t) * 16 + g * 4);
        x[2 * t] = x[2 * t] + f32x2{mk[0], mk[1]};
        x[2 * t + 1] = x[2 * t + 1] + f32x2{mk[2], mk[3]};
      }
    }
    float mx = -__builtin_huge_valf();
#pragma unroll
    for (int i = 0; i < P; ++i) mx = max_raw(mx, max_raw(x[i].x, x[i].y));
    mx = max_raw(mx, __shfl_xor(mx, 16));
    mx = max_raw(mx, __shfl_xor(mx, 32));
    if (SPLIT) mx = max_raw(mx, pair_exchange(s_red[0], mx, wave, kh, r16, g));   // (the barrier inside also publishes V^T)
    // The denominator's summation order is part of the contract (oracle/tq_int_oracle.c restates it): per key half,
    // a lane group adds its exponentials sequentially (tile-major), groups combine as (s0 + s1) + (s2 + s3), the two
    // halves are added last -- the same tree whether one wave owns the whole row or two waves own a half each.
    float sum = 0.f, sum_hi = 0.f;
    {
      const f32x2 nmx = {-mx, -mx};                    // x - mx == x + (-mx), bit for bit
#pragma unroll
      for (int i = 0; i < P; ++i) x[i] = x[i] + nmx;
    }
    exp_neg_ieee2_n<P>(x);
#pragma unroll
    for (int i = 0; i < P; ++i) {
      if (SPLIT || i < P / 2) { sum += x[i].x; sum += x[i].y; }
      else { sum_hi += x[i].x; sum_hi += x[i].y; }
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    if (SPLIT) {
      sum += pair_exchange(s_red[1], sum, wave, kh, r16, g);
    } else {
      sum_hi += __shfl_xor(sum_hi, 16);
      sum_hi += __shfl_xor(sum_hi, 32);
      sum += sum_hi;
    }
    bad_row = sum != sum;
    const float rsv = 1.0f / sum;                    // RN(e / sum), 1 <= sum <= T
    quot2_n<P>(x, f32x2{rsv, rsv}, f32x2{-sum, -sum});
    f32x2 hq[P];
#pragma unroll
    for (int c = 0; c < P; c += 4) {                 // clamp(rne(p / scale) + zp, lo, hi) - zp
      f32x2 (&xc)[4] = *reinterpret_cast<f32x2(*)[4]>(&x[c]);
      f32x2 (&hc)[4] = *reinterpret_cast<f32x2(*)[4]>(&hq[c]);
      qf_round2_n<4>(xc, fpq, hc);
    }
    const f32x2 zp2 = {pp.zp, pp.zp};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const f32x2 i0 = hq[2 * (s * 4 + tt)] + zp2, i1 = hq[2 * (s * 4 + tt) + 1] + zp2;     // indices in [0, 255]
        uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(i0.x, 0, 0u);
        w = __builtin_amdgcn_cvt_pk_u8_f32(i0.y, 1, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(i1.x, 2, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(i1.y, 3, w);
        fp[s][tt] = (int)(w ^ 0x80808080u);          // int8(index - 128)
      }
  } else {
    float mx = -__builtin_huge_valf();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 mk = {0.f, 0.f, 0.f, 0.f};
      if (p.mask) mk = *reinterpret_cast<const f32x4*>(p.mask + (size_t)b * T + (t0 + t) * 16 + g * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = sc[t][r];
        if (p.has_scores) {
          v = q_dequant(clamp_nanprop(rne_quot1(v, ps.scale, rcp_s) + ps.zp, ps.lo, ps.hi), ps);
        }
        v = denom_pow2 ? v * inv_denom : v / p.denom;      // x / 2^k == x * 2^-k exactly (sqrt(64) = 8)
        if (p.mask) v = v + mk[r];
        sc[t][r] = v;
        mx = max_raw(mx, v);
      }
    }
    // ---- softmax over the T keys of this lane's query: in-lane, then across the 4 lane groups ------------
    mx = max_raw(mx, __shfl_xor(mx, 16));
    mx = max_raw(mx, __shfl_xor(mx, 32));
    if (SPLIT) mx = max_raw(mx, pair_exchange(s_red[0], mx, wave, kh, r16, g));   // (the barrier inside also publishes V^T)
    float sum = 0.f, sum_hi = 0.f;                     // same summation tree as the branch-free form above
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sc[t][r] = exp_neg_ieee(sc[t][r] - mx);
        if (SPLIT || t < NT / 2) sum += sc[t][r];
        else sum_hi += sc[t][r];
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    if (SPLIT) {
      sum += pair_exchange(s_red[1], sum, wave, kh, r16, g);
    } else {
      sum_hi += __shfl_xor(sum_hi, 16);
      sum_hi += __shfl_xor(sum_hi, 32);
      sum += sum_hi;
    }
    bad_row = sum != sum;
    const float inv_sum = (sum >= 7.888609052210118e-31f && sum <= 1.2676506002282294e30f) ? 1.0f / sum : __builtin_nanf("");

    // ---- probability indices -> B operand of the second GEMM (byte tt * 4 + r of step s = key 64s + 16tt + 4g + r)
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        uint32_t word = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // rne((e / sum) / scale): e * (1/sum) * (1/scale) carries 4 roundings against the 2 of the exact chain
          // (<= 6 u apart); a band of 8 u (|h| + 1) around the ties decides which elements redo it exactly
          const float e = sc[s * 4 + tt][r];
          const float q0 = (e * inv_sum) * rcp_p;
          float hq = rintf(q0);
          if (!(fabsf(q0 - hq) < __builtin_fmaf(fabsf(hq), -2.0f * kTieTol, 0.5f - 2.0f * kTieTol)))   // 4 roundings: 8 u band
            hq = rintf((e / sum) / pp.scale);
          const int a = (int)clamp_nanprop(hq + pp.zp, pp.lo, pp.hi) - 128;
          word |= ((uint32_t)a & 0xffu) << (8 * r);
        }
        fp[s][tt] = (int)word;
      }
  }

  TQ_STAMP(4);
  if (!SPLIT && !K_LDS) __syncthreads();             // V^T is in LDS
  TQ_STAMP(5);

  // ---- C^T = V^T P^T -------------------------------------------------------------------------------
  // Key-split form (round 6): the kh = 1 wave is done once its probability indices are in LDS; its partner runs the second
  // GEMM over ALL keys.  (Before, both waves ran their half and the kh = 1 wave handed over 1 + 8 d / 16 integer partial
  // sums per lane -- 33 LDS stores, 33 loads and 33 additions per lane against 1 + 1 here; the MFMAs are not the
  // bottleneck.)  The same exact integers either way.
  v4i fpa[KSA];
#pragma unroll
  for (int s = 0; s < KS; ++s) fpa[s] = fp[s];
  if (SPLIT) {
    if (kh == 1) {
#pragma unroll
      for (int s = 0; s < KS; ++s) s_fp[wave][s][lane] = fp[s];
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int s = 0; s < KS; ++s) fpa[KS + s] = s_fp[wave][s][lane];
  }
  v4i rsp4 = zero4;
#pragma unroll
  for (int s = 0; s < KSA; ++s) rsp4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ones, fpa[s], rsp4, 0, 0, 0);
  v4i accs[DH / 16], csvs[DH / 16];
#pragma unroll
  for (int j = 0; j < DH / 16; ++j) {
    v4i acc = zero4, csv = zero4;
#pragma unroll
    for (int s = 0; s < KSA; ++s) {
      const v4i fv = *reinterpret_cast<const v4i*>(s_vt + (j * 16 + r16) * PITCH + s * 64 + g * 16);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(fv, fpa[s], acc, 0, 0, 0);
      csv = __builtin_amdgcn_mfma_i32_16x16x64_i8(fv, ones, csv, 0, 0, 0);              // sum_k a'_v of rows 4g + r
    }
    accs[j] = acc;
    csvs[j] = csv;
  }
  const int rsp = rsp4[0];                           // sum_k a'_p of column r16
  const int p_const = cv * rsp + T * cp * cv;
  const size_t out_row = ((size_t)b * T + qrow) * ((size_t)p.H * DH) + (size_t)h * DH;
#pragma unroll
  for (int j = 0; j < DH / 16; ++j) {
    const v4i acc = accs[j], csv = csvs[j];
    float o[4];
    uint32_t oi = 0;
    if (fast_ctx) {                                  // exact branch-free quantizer (tq_device.h QF): no IEEE division
      f32x2 v2[2], h2[2];
#pragma unroll
      for (int r = 0; r < 4; ++r) v2[r >> 1][r & 1] = (float)(acc[r] + cp * csv[r] + p_const) * s_pv;
      qf_round2_n<2>(v2, fc, h2);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hh = h2[r >> 1][r & 1];
        oi = __builtin_amdgcn_cvt_pk_u8_f32(hh + pc.zp, r, oi);
        o[r] = bad_row ? __builtin_nanf("") : pc.scale * (hh + 0.0f);
      }
      oi ^= 0x80808080u;                             // int8(index - 128)
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = (float)(acc[r] + cp * csv[r] + p_const) * s_pv;
        if (p.has_ctx) {
          const float xi = q_index(v, pc);
          oi |= ((uint32_t)((int)xi - 128) & 0xffu) << (8 * r);
          v = q_dequant(xi, pc);
        }
        o[r] = bad_row ? __builtin_nanf("") : v;       // softmax of a fully masked row is NaN in the reference
      }
    }
    const size_t off = out_row + j * 16 + g * 4;
    *reinterpret_cast<f32x4*>(p.ctx + off) = f32x4{o[0], o[1], o[2], o[3]};
    if (p.ctx_idx) *reinterpret_cast<uint32_t*>(p.ctx_idx + off) = oi;
  }
  TQ_STAMP(6);
}

}  // namespace tq

using namespace tq;

static int check_i8_grid(const tq_quantizer* q, const char* what) {
  TQ_REQUIRE(q != nullptr && q->delta != nullptr && q->zero_float != nullptr, "tq_attention_i8_fwd: %s quantizer missing", what);
  TQ_REQUIRE(!q->symmetric && !q->log_domain && q->n_params == 1 && q->n_bits >= 1 && q->n_bits <= 8,
             "tq_attention_i8_fwd: %s must be a per-tensor asymmetric linear-domain quantizer with n_bits <= 8", what);
  return TQ_OK;
}

extern "C" int tq_attention_i8_fwd(const int8_t* q_idx, const int8_t* k_idx, const int8_t* v_idx, float* ctx,
                                   int8_t* ctx_idx, uint64_t B, uint64_t T, uint64_t H, uint64_t head_dim,
                                   uint64_t qkv_row_stride, const float* mask, float denom, const tq_quantizer* q_q,
                                   const tq_quantizer* q_k, const tq_quantizer* q_v, const tq_quantizer* q_scores,
                                   const tq_quantizer* q_probs, const tq_quantizer* q_ctx, tq_stream_t stream) {
  return tq_attention_i8_strided_fwd(q_idx, k_idx, v_idx, ctx, ctx_idx, B, T, H, head_dim, qkv_row_stride, qkv_row_stride, mask,
                                     denom, q_q, q_k, q_v, q_scores, q_probs, q_ctx, stream);
}

extern "C" int tq_attention_i8_strided_fwd(const int8_t* q_idx, const int8_t* k_idx, const int8_t* v_idx, float* ctx,
                                           int8_t* ctx_idx, uint64_t B, uint64_t T, uint64_t H, uint64_t head_dim,
                                           uint64_t qkv_row_stride, uint64_t v_row_stride, const float* mask, float denom,
                                           const tq_quantizer* q_q, const tq_quantizer* q_k, const tq_quantizer* q_v,
                                           const tq_quantizer* q_scores, const tq_quantizer* q_probs,
                                           const tq_quantizer* q_ctx, tq_stream_t stream) {
  if (B == 0 || T == 0 || H == 0) return TQ_OK;
  TQ_REQUIRE(q_idx && k_idx && v_idx && ctx, "tq_attention_i8_fwd: NULL pointer");
  TQ_REQUIRE(head_dim == 64 || head_dim == 32, "tq_attention_i8_fwd: head_dim %llu unsupported (32, 64)", (unsigned long long)head_dim);
  TQ_REQUIRE(T % 64 == 0 && T <= 512, "tq_attention_i8_fwd: sequence length %llu unsupported (multiples of 64 up to 512)",
             (unsigned long long)T);
  TQ_REQUIRE(aligned16(q_idx) && aligned16(k_idx) && aligned16(v_idx) && aligned16(ctx) &&
             (mask == nullptr || aligned16(mask)) && (ctx_idx == nullptr || (reinterpret_cast<uintptr_t>(ctx_idx) % 4) == 0),
             "tq_attention_i8_fwd: 16-byte alignment required");
  TQ_REQUIRE(denom != 0.0f, "tq_attention_i8_fwd: denom == 0");
  if (qkv_row_stride == 0) qkv_row_stride = H * head_dim;
  if (v_row_stride == 0) v_row_stride = H * head_dim;
  TQ_REQUIRE(qkv_row_stride >= H * head_dim && qkv_row_stride % 16 == 0 && qkv_row_stride < (1ull << 31),
             "tq_attention_i8_fwd: bad qkv_row_stride %llu", (unsigned long long)qkv_row_stride);
  TQ_REQUIRE(v_row_stride >= H * head_dim && v_row_stride % 16 == 0 && v_row_stride < (1ull << 31),
             "tq_attention_i8_fwd: bad v_row_stride %llu", (unsigned long long)v_row_stride);
  TQ_REQUIRE(B * H * (T / (16 * kAttnWaves)) < (1ull << 31), "tq_attention_i8_fwd: too many tiles");
  if (int e = check_i8_grid(q_q, "query")) return e;
  if (int e = check_i8_grid(q_k, "key")) return e;
  if (int e = check_i8_grid(q_v, "value")) return e;
  if (int e = check_i8_grid(q_probs, "probabilities")) return e;
  if (q_scores) {
    if (int e = check_quantizer(q_scores, B * H * T * T, "tq_attention_i8_fwd")) return e;
    TQ_REQUIRE(q_scores->n_params == 1, "tq_attention_i8_fwd: per-tensor score quantizer only");
  }
  if (q_ctx) {
    if (int e = check_quantizer(q_ctx, B * T * H * head_dim, "tq_attention_i8_fwd")) return e;
    TQ_REQUIRE(q_ctx->n_params == 1, "tq_attention_i8_fwd: per-tensor context quantizer only");
    TQ_REQUIRE(ctx_idx == nullptr || (!q_ctx->symmetric && q_ctx->n_bits <= 8),
               "tq_attention_i8_fwd: ctx_idx needs an asymmetric <= 8-bit context quantizer");
  } else {
    TQ_REQUIRE(ctx_idx == nullptr, "tq_attention_i8_fwd: ctx_idx needs q_ctx");
  }
  AttnArgs a{};
  a.q = q_idx; a.k = k_idx; a.v = v_idx; a.ctx = ctx; a.ctx_idx = ctx_idx; a.mask = mask;
  a.B = (uint32_t)B; a.T = (uint32_t)T; a.H = (uint32_t)H; a.in_stride = (uint32_t)qkv_row_stride; a.v_stride = (uint32_t)v_row_stride; a.denom = denom;
  a.qq = *q_q; a.qk = *q_k; a.qv = *q_v; a.q_probs = *q_probs;
  a.has_scores = q_scores != nullptr; a.has_ctx = q_ctx != nullptr;
  a.fast_ok = tuning("TQ_ATTN_FAST", 1);
#ifdef TQ_ATTN_PROF
  { const char* e = getenv("TQ_ATTN_PROF_PTR"); a.prof = e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0)) : nullptr; }
#endif
  if (q_scores) a.q_scores = *q_scores;
  if (q_ctx) a.q_ctx = *q_ctx;
  const unsigned grid = (unsigned)(B * H * (T / (16 * kAttnWaves)));
  hipStream_t st = static_cast<hipStream_t>(stream);
  // key split (4 waves per 32 queries) while the plain grid leaves the SIMDs under ~2 waves each: T % 128 == 0 only
  const int split_env = tuning("TQ_ATTN_SPLIT", -1);     // read per call: the tests flip it
  const bool split = T % 128 == 0 && (split_env >= 0 ? split_env != 0 : grid <= 1024);
  // eight query waves per workgroup (one V^T staging per 128 queries) whenever the keys are not split: T % 128 == 0 only
  // (profiles/r06/attn_qw_ab.txt: -3 % at B = 32, -7 % at B = 64, -9 % at B = 128, -10 % at T = 256, -20 % at T = 512)
  const int qw_env = tuning("TQ_ATTN_QW", -1);           // read per call: the tests flip it
  const bool wide = !split && T % 128 == 0 && (qw_env >= 0 ? qw_env == 8 : true);
#define TQ_ATTN_LAUNCH(NTV, DHV, SP)                                                                             \
  hipLaunchKernelGGL((attention_i8_k<NTV, DHV, SP>), dim3(grid), dim3((SP) ? 2 * kAttnThreads : kAttnThreads), 0, st, a)
#define TQ_ATTN_LAUNCH8(NTV, DHV)                                                                                \
  hipLaunchKernelGGL((attention_i8_k<NTV, DHV, false, 8>), dim3(grid / 4), dim3(8 * kWave), 0, st, a)
#define TQ_ATTN(NTV)                                                                                             \
  case NTV * 16:                                                                                                 \
    if constexpr ((NTV) % 8 == 0) {                                                                              \
      if (split) {                                                                                               \
        if (head_dim == 64) TQ_ATTN_LAUNCH(NTV, 64, true);                                                       \
        else TQ_ATTN_LAUNCH(NTV, 32, true);                                                                      \
        break;                                                                                                   \
      }                                                                                                          \
    }                                                                                                            \
    if constexpr ((NTV) % 8 == 0) {                                                                              \
      if (wide) {                                                                                                \
        if (head_dim == 64) TQ_ATTN_LAUNCH8(NTV, 64);                                                            \
        else TQ_ATTN_LAUNCH8(NTV, 32);                                                                           \
        break;                                                                                                   \
      }                                                                                                          \
    }                                                                                                            \
    if (head_dim == 64) TQ_ATTN_LAUNCH(NTV, 64, false);                                                          \
    else TQ_ATTN_LAUNCH(NTV, 32, false);                                                                         \
    break
  switch (T) {
    TQ_ATTN(4); TQ_ATTN(8); TQ_ATTN(12); TQ_ATTN(16); TQ_ATTN(20); TQ_ATTN(24); TQ_ATTN(28); TQ_ATTN(32);
    default: return set_error(TQ_EUNSUPPORTED, "tq_attention_i8_fwd: sequence length %llu", (unsigned long long)T);
  }
#undef TQ_ATTN_LAUNCH
#undef TQ_ATTN_LAUNCH8
#undef TQ_ATTN
  return check_launch("attention_i8_k");
}
