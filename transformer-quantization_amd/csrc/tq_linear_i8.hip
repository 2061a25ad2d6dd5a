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
// Kernel shape: LDS-free.  A wave owns a TN x TM output tile (TN, TM in {32, 64}); per 64-byte K step
// every lane loads 16 contiguous bytes of each of its TN/16 + TM/16 operand rows straight into the
// MFMA fragment layout (lane l: row l & 15, k bytes (l >> 4) * 16 ...; any k permutation is legal as
// long as both operands use the same one), next step's loads are issued before this step's MFMAs.
// The operands are tiny (<= 3 MB) and L2-resident; the 4 waves of a block share their W rows.
// The W tile is the FIRST MFMA operand so that a lane's 4 accumulator registers are 4 consecutive
// output features of one token: 16-byte stores.
#include <algorithm>

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
  tq_quantizer q_out;
};

template <int TN, int TM, int YDT>
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

  // ---- epilogue: zero-point correction, scales, bias, activation, output quantizer ----------------
  const float dx = p.x_delta[0];
  const float sx = dx < p.x_eps ? p.x_eps : dx;
  const int zx = (int)clamp_nanprop(rintf(p.x_zero_float[0]), 0.0f, grid_top(p.x_n_bits));
  const int shift = 128 - zx;
  QP qo = {1.f, 0.f, 0.f, 0.f};
  if (p.has_q) qo = make_qp(p.q_out, 0);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const uint32_t n = n0 + i * 16 + kg * 4;            // this lane's 4 consecutive output features
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
      const uint32_t m = m0 + j * 16 + r16;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = (float)(acc[i][j][r] + rs[r]) * sw[r] + bs[r];
        v = apply_act(v, p.act);
        if (p.has_q) v = q_dequant(q_index(v, qo), qo);
        o[r] = v;
      }
      if (YDT == TQ_F32) {
        *reinterpret_cast<f32x4*>(static_cast<float*>(p.y) + (size_t)m * p.N + n) = f32x4{o[0], o[1], o[2], o[3]};
      } else {
        u32x2 pk;
        f32x2 a = {o[0], o[1]}, b = {o[2], o[3]};
        pk[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2));
        pk[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, bf16x2));
        *reinterpret_cast<u32x2*>(static_cast<uint16_t*>(p.y) + (size_t)m * p.N + n) = pk;
      }
    }
  }
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

template <int YDT>
static int launch_linear(const LinArgs& a, hipStream_t st) {
  // prefer 64x64 wave tiles when they still give >= 512 waves, else 32x32
  const bool big = (a.M % 64 == 0) && (a.N % 64 == 0) && ((uint64_t)(a.M / 64) * (a.N / 64) >= 512);
  if (big) {
    const unsigned tiles = (a.M / 64) * (a.N / 64);
    hipLaunchKernelGGL((linear_i8_k<64, 64, YDT>), dim3((unsigned)ceil_div(tiles, kBlock / kWave)), dim3(kBlock), 0, st, a);
  } else {
    const unsigned tiles = (a.M / 32) * (a.N / 32);
    hipLaunchKernelGGL((linear_i8_k<32, 32, YDT>), dim3((unsigned)ceil_div(tiles, kBlock / kWave)), dim3(kBlock), 0, st, a);
  }
  return check_launch("linear_i8_k");
}

}  // namespace tq

using namespace tq;

extern "C" int tq_rowsum_i8(const int8_t* w_idx, int32_t* rowsum, uint64_t N, uint64_t K, tq_stream_t stream) {
  TQ_REQUIRE(w_idx && rowsum && N >= 1 && K >= 1, "tq_rowsum_i8: bad argument");
  hipLaunchKernelGGL(rowsum_i8_k, dim3((unsigned)ceil_div(N, kBlock / kWave)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), w_idx, rowsum, (uint32_t)N, (uint32_t)K);
  return check_launch("rowsum_i8_k");
}

extern "C" int tq_linear_i8_fwd(const int8_t* x_idx, const int8_t* w_idx, const int32_t* w_rowsum, const float* bias,
                                void* y, int y_dtype, uint64_t M, uint64_t N, uint64_t K, const float* x_delta,
                                const float* x_zero_float, int x_n_bits, float x_eps, const float* w_delta,
                                uint64_t w_n_params, float w_eps, int activation, const tq_quantizer* q_out,
                                tq_stream_t stream) {
  if (M == 0 || N == 0) return TQ_OK;
  TQ_REQUIRE(x_idx && w_idx && w_rowsum && y && x_delta && x_zero_float && w_delta, "tq_linear_i8_fwd: NULL pointer");
  TQ_REQUIRE(y_dtype == TQ_F32 || y_dtype == TQ_BF16, "tq_linear_i8_fwd: y dtype must be fp32 or bf16");
  TQ_REQUIRE(M % 32 == 0 && N % 32 == 0 && K % 64 == 0 && K >= 64 && K <= 16384 && M < (1u << 31) && N < (1u << 31),
             "tq_linear_i8_fwd: unsupported shape M=%llu N=%llu K=%llu (M,N %% 32, K %% 64)", (unsigned long long)M,
             (unsigned long long)N, (unsigned long long)K);
  TQ_REQUIRE(x_n_bits >= 1 && x_n_bits <= 8, "tq_linear_i8_fwd: input quantizer must have <= 8 bits");
  TQ_REQUIRE(w_n_params == 1 || w_n_params == N, "tq_linear_i8_fwd: weight scales must be per-tensor or per-output-channel");
  TQ_REQUIRE(activation >= ACT_NONE && activation <= ACT_TANH, "tq_linear_i8_fwd: unknown activation %d", activation);
  TQ_REQUIRE(aligned16(x_idx) && aligned16(w_idx) && aligned16(y), "tq_linear_i8_fwd: 16-byte alignment required");
  LinArgs a{};
  a.x = x_idx; a.w = w_idx; a.w_rowsum = w_rowsum; a.bias = bias; a.y = y;
  a.M = (uint32_t)M; a.N = (uint32_t)N; a.K = (uint32_t)K;
  a.x_delta = x_delta; a.x_zero_float = x_zero_float; a.x_eps = x_eps; a.x_n_bits = x_n_bits;
  a.w_delta = w_delta; a.w_n_params = (uint32_t)w_n_params; a.w_eps = w_eps; a.act = activation;
  a.has_q = q_out != nullptr;
  if (q_out) {
    if (int e = check_quantizer(q_out, M * N, "tq_linear_i8_fwd")) return e;
    TQ_REQUIRE(q_out->n_params == 1, "tq_linear_i8_fwd: per-tensor output quantizer only");
    a.q_out = *q_out;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  return y_dtype == TQ_F32 ? launch_linear<TQ_F32>(a, st) : launch_linear<TQ_BF16>(a, st);
}
