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
// One workgroup (2 waves) owns 32 query rows of one (batch, head); each wave computes S^T = K Q^T for
// its 16 queries with T/16 v_mfma_i32_16x16x64_i8 (d = 64 = one MFMA K step), keeps the 16 x T scores
// in registers (lane = one query column, keys 16t + 4g + r), does the quantizers / mask / softmax
// there, packs the probability indices straight into the B operand of the second MFMA (the MFMA K
// dimension may be permuted freely as long as both operands agree, so the accumulator layout IS the
// operand layout once V^T is stored with the matching key permutation in LDS) and computes
// C^T = V^T P^T.  Row / column sums for the zero-point corrections are MFMAs against an all-ones
// operand.  Nothing but Q, K, V indices (3 x 64 B per token and head) is read and only C is written:
// the [B, H, T, T] score and probability tensors never exist in memory.
#include <algorithm>

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
  uint32_t in_stride;         // elements between consecutive tokens of q / k / v (H * 64, or 3 * H * 64 inside a stacked QKV buffer)
  float denom;
  tq_quantizer qq, qk, qv;    // per-tensor asymmetric, n_bits <= 8
  tq_quantizer q_scores, q_probs, q_ctx;
  int has_scores, has_ctx;
};

// position of key (64 s + 16 tt + 4 g + r) inside a V^T row: 64 s + 16 g + 4 tt + r
__device__ __forceinline__ uint32_t key_slot(uint32_t key) {
  return (key & ~63u) | (((key >> 2) & 3u) << 4) | (((key >> 4) & 3u) << 2) | (key & 3u);
}

template <int NT, int DH>   // NT = T / 16 key tiles, DH = head dim (32 or 64)
__global__ __launch_bounds__(kAttnThreads) void attention_i8_k(AttnArgs p) {   // up to 512 VGPRs per lane at 2 waves
  constexpr int T = NT * 16, KS = NT / 4;          // KS = 64-key MFMA steps of the second GEMM
  constexpr int PITCH = T + 32;                    // 32 * odd bytes: conflict-free ds_read_b128 (4 x 16 lane groups, 64 banks)
  __shared__ __attribute__((aligned(16))) int8_t s_vt[DH * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const uint32_t qblocks = T / (16 * kAttnWaves);
  const uint32_t bh = blockIdx.x / qblocks, qb = blockIdx.x % qblocks;
  const uint32_t b = bh / p.H, h = bh % p.H;
  const size_t row_stride = p.in_stride;
  const size_t base = (size_t)b * T * row_stride + (size_t)h * DH;

  // ---- V^T -> LDS with the key permutation of the accumulator layout -------------------------------
  // One work item = 4 consecutive keys x 16 head dims: four 16-byte loads, 4x4 byte transposes in
  // registers, sixteen 32-bit LDS stores (keys 4m .. 4m+3 are adjacent slots of one V^T row).
  constexpr uint32_t PARTS = DH / 16;
  for (uint32_t c = tid; c < (uint32_t)T / 4 * PARTS; c += kAttnThreads) {
    const uint32_t key4 = (c / PARTS) * 4, part = c % PARTS;
    v4i raw[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      raw[kk] = *reinterpret_cast<const v4i*>(p.v + base + (size_t)(key4 + kk) * row_stride + part * 16);
    const uint32_t slot = key_slot(key4);
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint32_t word = 0;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) word |= (((uint32_t)raw[kk][w] >> (8 * e)) & 0xffu) << (8 * kk);
        *reinterpret_cast<uint32_t*>(s_vt + (part * 16 + w * 4 + e) * PITCH + slot) = word;
      }
  }

  // ---- S^T = K Q^T for this wave's 16 queries --------------------------------------------------------
  const v4i ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
  const v4i zero4 = {0, 0, 0, 0};
  const uint32_t qrow = qb * 16 * kAttnWaves + wave * 16 + r16;
  const bool kin = g * 16 < DH;                     // lane groups beyond the head dim supply zeros
  v4i fq = zero4;
  if (kin) fq = *reinterpret_cast<const v4i*>(p.q + base + (size_t)qrow * row_stride + g * 16);

  const QP pq = make_qp(p.qq, 0), pk = make_qp(p.qk, 0), pv = make_qp(p.qv, 0), pp = make_qp(p.q_probs, 0);
  const int cq = 128 - (int)pq.zp, ck = 128 - (int)pk.zp, cv = 128 - (int)pv.zp, cp = 128 - (int)pp.zp;
  const float s_qk = pq.scale * pk.scale, s_pv = pp.scale * pv.scale;
  QP ps = {1.f, 0.f, 0.f, 0.f}, pc = {1.f, 0.f, 0.f, 0.f};
  if (p.has_scores) ps = make_qp(p.q_scores, 0);
  if (p.has_ctx) pc = make_qp(p.q_ctx, 0);
  const float rcp_s = guarded_rcp(ps.scale), rcp_p = guarded_rcp(pp.scale);   // rne(x / scale), tq_device.h

  // denom = 2^k (normal range): the division is an exact scaling
  const uint32_t dbits = f32_to_bits(p.denom);
  const bool denom_pow2 = (dbits & 0x007fffffu) == 0 && (dbits >> 23) >= 32 && (dbits >> 23) <= 222;
  const float inv_denom = 1.0f / p.denom;

  const int rsq = __builtin_amdgcn_mfma_i32_16x16x64_i8(ones, fq, zero4, 0, 0, 0)[0];   // sum_d a'_q of column r16
  const int q_const = ck * rsq + DH * cq * ck;

  float sc[NT][4];
  float mx = -__builtin_huge_valf();
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    v4i fk = zero4;
    if (kin) fk = *reinterpret_cast<const v4i*>(p.k + base + (size_t)(t * 16 + r16) * row_stride + g * 16);
    const v4i acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(fk, fq, zero4, 0, 0, 0);
    const v4i rsk = __builtin_amdgcn_mfma_i32_16x16x64_i8(fk, ones, zero4, 0, 0, 0);    // sum_d a'_k of rows 4g + r
    f32x4 mk = {0.f, 0.f, 0.f, 0.f};
    if (p.mask) mk = *reinterpret_cast<const f32x4*>(p.mask + (size_t)b * T + t * 16 + g * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = (float)(acc[r] + cq * rsk[r] + q_const) * s_qk;
      if (p.has_scores) {
        v = q_dequant(clamp_nanprop(rne_quot1(v, ps.scale, rcp_s) + ps.zp, ps.lo, ps.hi), ps);
      }
      v = denom_pow2 ? v * inv_denom : v / p.denom;      // x / 2^k == x * 2^-k exactly (sqrt(64) = 8)
      if (p.mask) v = v + mk[r];
      sc[t][r] = v;
      mx = fmaxf(mx, v);
    }
  }
  // ---- softmax over the T keys of this lane's query: in-lane, then across the 4 lane groups ------------
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { sc[t][r] = expf(sc[t][r] - mx); sum += sc[t][r]; }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float inv_sum = (sum >= 7.888609052210118e-31f && sum <= 1.2676506002282294e30f) ? 1.0f / sum : __builtin_nanf("");

  // ---- probability indices -> B operand of the second GEMM (byte tt * 4 + r of step s = key 64s + 16tt + 4g + r)
  v4i fp[KS];
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

  __syncthreads();                                   // V^T is in LDS

  // ---- C^T = V^T P^T -------------------------------------------------------------------------------
  v4i rsp4 = zero4;
#pragma unroll
  for (int s = 0; s < KS; ++s) rsp4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ones, fp[s], rsp4, 0, 0, 0);
  const int p_const = cv * rsp4[0] + T * cp * cv;    // sum_k a'_p of column r16
  const size_t out_row = ((size_t)b * T + qrow) * ((size_t)p.H * DH) + (size_t)h * DH;
#pragma unroll
  for (int j = 0; j < DH / 16; ++j) {
    v4i acc = zero4, csv = zero4;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const v4i fv = *reinterpret_cast<const v4i*>(s_vt + (j * 16 + r16) * PITCH + s * 64 + g * 16);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(fv, fp[s], acc, 0, 0, 0);
      csv = __builtin_amdgcn_mfma_i32_16x16x64_i8(fv, ones, csv, 0, 0, 0);              // sum_k a'_v of rows 4g + r
    }
    float o[4];
    struct alignas(4) { int8_t e[4]; } oi = {{0, 0, 0, 0}};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = (float)(acc[r] + cp * csv[r] + p_const) * s_pv;
      if (p.has_ctx) {
        const float xi = q_index(v, pc);
        oi.e[r] = (int8_t)((int)xi - 128);
        v = q_dequant(xi, pc);
      }
      o[r] = v;
    }
    const size_t off = out_row + j * 16 + g * 4;
    *reinterpret_cast<f32x4*>(p.ctx + off) = f32x4{o[0], o[1], o[2], o[3]};
    if (p.ctx_idx) *reinterpret_cast<uint32_t*>(p.ctx_idx + off) = __builtin_bit_cast(uint32_t, oi);
  }
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
  TQ_REQUIRE(qkv_row_stride >= H * head_dim && qkv_row_stride % 16 == 0 && qkv_row_stride < (1ull << 31),
             "tq_attention_i8_fwd: bad qkv_row_stride %llu", (unsigned long long)qkv_row_stride);
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
  a.B = (uint32_t)B; a.T = (uint32_t)T; a.H = (uint32_t)H; a.in_stride = (uint32_t)qkv_row_stride; a.denom = denom;
  a.qq = *q_q; a.qk = *q_k; a.qv = *q_v; a.q_probs = *q_probs;
  a.has_scores = q_scores != nullptr; a.has_ctx = q_ctx != nullptr;
  if (q_scores) a.q_scores = *q_scores;
  if (q_ctx) a.q_ctx = *q_ctx;
  const unsigned grid = (unsigned)(B * H * (T / (16 * kAttnWaves)));
  hipStream_t st = static_cast<hipStream_t>(stream);
#define TQ_ATTN(NTV)                                                                                             \
  case NTV * 16:                                                                                                 \
    if (head_dim == 64) hipLaunchKernelGGL((attention_i8_k<NTV, 64>), dim3(grid), dim3(kAttnThreads), 0, st, a); \
    else hipLaunchKernelGGL((attention_i8_k<NTV, 32>), dim3(grid), dim3(kAttnThreads), 0, st, a);                \
    break
  switch (T) {
    TQ_ATTN(4); TQ_ATTN(8); TQ_ATTN(12); TQ_ATTN(16); TQ_ATTN(20); TQ_ATTN(24); TQ_ATTN(28); TQ_ATTN(32);
    default: return set_error(TQ_EUNSUPPORTED, "tq_attention_i8_fwd: sequence length %llu", (unsigned long long)T);
  }
#undef TQ_ATTN
  return check_launch("attention_i8_k");
}
