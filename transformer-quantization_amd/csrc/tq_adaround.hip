// K10/K11/K13: AdaRound kernels for gfx950 (fp32 weights, like the reference).
//
//   ada_fwd_k      w_q = s * (clamp(floor(w/s) + r (+zp), lo, hi) - zp), r = h(alpha) | [alpha >= 0]
//                  (reference adaround/quantizer.py:46-90 + quantizers.py:209): ~8 ATen kernels -> 1
//   ada_init_k     alpha such that h(alpha) = frac(w/s)  (adaround/quantizer.py:57-71)
//   ada_bwd_adam_k autograd through ada_fwd + rounding regulariser (adaround/utils.py:159-162)
//                  + torch.optim.Adam step on alpha (adaround/adaround.py:98-99, 260): ~25 -> 1,
//                  28 B/element (read g, w, alpha, m, v; write alpha, m, v)
//   ada_reg_k      value of the regulariser; recon_k: mse(pred,tgt,'none').sum(1).mean()
#include <algorithm>

#include "tq_device.h"
#include "tq_host.h"

namespace tq {

// python-double constants of adaround/quantizer.py:27-34 as ATen narrows them for fp32 tensors
constexpr float kZeta = 1.1f;
constexpr float kGamma = -0.1f;
constexpr float kStretch = (float)(1.1 - (-0.1));   // zeta - gamma evaluated in double, then fp32

__device__ __forceinline__ float sigmoidf_(float a) { return 1.0f / (1.0f + expf(-a)); }

// ---- hardware-rate transcendentals for the per-iteration kernels (K10 soft forward, K11 fused step) ----------------
// The optimisation loop's relaxation h(alpha), its derivative, the regulariser gradient and the Adam update are
// compared with the reference at a tolerance (they run through torch's vectorised CPU exp / pow there, which no GPU
// kernel matches bit for bit anyway); what IS contractual -- floor(w / s), the clamp, the hard rounding [alpha >= 0] --
// stays exact below.  v_exp_f32 / v_log_f32 / v_rcp_f32 / v_sqrt_f32 are 1-ulp, quarter-rate instructions: a sigmoid is
// ~10 issue slots instead of ~26 (expf + IEEE division), |2h-1|^(beta-1) ~10 instead of ~60 (powf), the Adam quotient
// ~11 instead of ~30 (IEEE sqrt + two IEEE divisions).  K11 drops from ~390 to ~80 issue slots per element and becomes
// what SURVEY 8(d) says it is: a 28 B/element streaming kernel.
constexpr float kLog2e = 1.44269504088896340736f;
__device__ __forceinline__ float fast_sigmoid(float a) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a * -kLog2e));
}
// u^e for u >= 0 (u = 0 -> 0 for e > 0; callers handle e == 0)
__device__ __forceinline__ float fast_pow(float u, float e) { return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(u)); }

// h(alpha) and dh/dalpha with the fast sigmoid; inv_temp = 1 / temp.  MODE is a template parameter of the
// per-iteration kernels: the element body is then branch-free and the V element chains of a thread interleave.
template <int MODE>
__device__ __forceinline__ float ada_h_fast(float a, float inv_temp, float& dh) {
  if (MODE == TQ_ADA_HARD_SIGMOID) {
    const float s = fast_sigmoid(a);
    const float u = s * kStretch + kGamma;
    dh = (u >= 0.0f && u <= 1.0f) ? kStretch * (s * (1.0f - s)) : 0.0f;
    return clamp_nanprop(u, 0.0f, 1.0f);
  }
  if (MODE == TQ_ADA_SIGMOID) {
    const float s = fast_sigmoid(a);
    dh = s * (1.0f - s);
    return s;
  }
  const float s = fast_sigmoid(a * inv_temp);
  dh = (s * (1.0f - s)) * inv_temp;
  return s;
}
__device__ __forceinline__ float ada_h_fast(float a, int mode, float inv_temp, float& dh) {
  if (mode == TQ_ADA_HARD_SIGMOID) return ada_h_fast<TQ_ADA_HARD_SIGMOID>(a, inv_temp, dh);
  if (mode == TQ_ADA_SIGMOID) return ada_h_fast<TQ_ADA_SIGMOID>(a, inv_temp, dh);
  return ada_h_fast<TQ_ADA_SIGMOID_TEMP>(a, inv_temp, dh);
}

// floor(RN(w / scale)) without the IEEE division: Markstein's correctly rounded quotient from the correctly rounded
// reciprocal r = RN(1 / scale) (tq_device.h, identity (1); exact-rational proof in tests/test_exact_quotient.py), with
// the true division for the operands the identity does not cover (huge / non-finite quotients, scales outside
// [2^-100, 2^100]: rcp = NaN makes the comparison fail).  Bit-identical to floorf(w / scale) for every input.
__device__ __forceinline__ float floor_quot(float w, float scale, float rcp) {
  const float q0 = w * rcp;
  float q1 = __builtin_fmaf(__builtin_fmaf(-q0, scale, w), rcp, q0);
  if (!(fabsf(q0) < 4194304.0f)) q1 = w / scale;
  return floorf(q1);
}
// V quotients of one parameter: the Markstein chains side by side, ONE (practically never taken) branch per vector for
// the operands outside the identity's domain instead of one per element
template <int V, class VEC>
__device__ __forceinline__ void floor_quot_n(const VEC& w, float scale, float rcp, float (&fl)[V]) {
  float q0[V], q1[V];
  bool odd = false;
#pragma unroll
  for (int j = 0; j < V; ++j) q0[j] = w[j] * rcp;
#pragma unroll
  for (int j = 0; j < V; ++j) q1[j] = __builtin_fmaf(-q0[j], scale, w[j]);
#pragma unroll
  for (int j = 0; j < V; ++j) {
    q1[j] = __builtin_fmaf(q1[j], rcp, q0[j]);
    odd = odd || !(fabsf(q0[j]) < 4194304.0f);
  }
  if (__builtin_expect(odd, 0)) {
#pragma unroll
    for (int j = 0; j < V; ++j)
      if (!(fabsf(q0[j]) < 4194304.0f)) q1[j] = w[j] / scale;
  }
#pragma unroll
  for (int j = 0; j < V; ++j) fl[j] = floorf(q1[j]);
}

// h(alpha) and dh/dalpha
__device__ __forceinline__ float ada_h(float a, int mode, float temp, float* dh) {
  if (mode == TQ_ADA_SIGMOID) {
    const float s = sigmoidf_(a);
    if (dh) *dh = s * (1.0f - s);
    return s;
  }
  if (mode == TQ_ADA_HARD_SIGMOID) {
    const float s = sigmoidf_(a);
    const float u = s * kStretch + kGamma;
    if (dh) *dh = (u >= 0.0f && u <= 1.0f) ? kStretch * (s * (1.0f - s)) : 0.0f;   // clamp mask is inclusive
    return clamp_nanprop(u, 0.0f, 1.0f);
  }
  const float s = sigmoidf_(a / temp);
  if (dh) *dh = (s * (1.0f - s)) / temp;
  return s;
}

__device__ __forceinline__ uint64_t par_index(const tq_quantizer& q, uint64_t i) {
  return q.n_params == 1 ? 0 : (i / q.inner) % q.n_params;
}

// V consecutive elements per thread (16-byte accesses, one parameter look-up per vector); V = 1 for ragged / unaligned /
// per-channel tensors whose inner extent is not a multiple of 4
template <int V>
struct AdaVec { typedef float type __attribute__((ext_vector_type(V))); };

template <int V, int MODE, bool SOFT>
__global__ __launch_bounds__(kBlock) void ada_fwd_k(const float* __restrict__ w, const float* __restrict__ alpha,
                                                    float* __restrict__ out, uint64_t n, tq_quantizer q, float temp) {
  typedef typename AdaVec<V>::type vec;
  const uint64_t nv = n / V;
  const float inv_temp = MODE == TQ_ADA_SIGMOID_TEMP ? 1.0f / temp : 1.0f;
  const bool one = q.n_params == 1;
  // (make_qp's select form in this file: the parameters are also derived per vector inside the element loops, where the
  // volatile asm of the branch form blocks hoisting; one box, ada_bwd_adam_k [30522,768]: 128.6 us against 136.1 us,
  // profiles/r06/header_ab_same_box.txt)
  QP p = make_qp<true>(q, 0);                              // per-tensor grid: loop-invariant
  float rcp = guarded_rcp(p.scale);
  for (uint64_t iv = (uint64_t)blockIdx.x * kBlock + threadIdx.x; iv < nv; iv += (uint64_t)gridDim.x * kBlock) {
    if (!one) {
      p = make_qp<true>(q, par_index(q, iv * V));
      rcp = guarded_rcp(p.scale);
    }
    const vec wv = reinterpret_cast<const vec*>(w)[iv], av = reinterpret_cast<const vec*>(alpha)[iv];
    float fl[V];
    floor_quot_n<V>(wv, p.scale, rcp, fl);
    vec o;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float a = av[j];
      float dh_unused;
      const float r = SOFT ? ada_h_fast<MODE>(a, inv_temp, dh_unused) : (a >= 0.0f ? 1.0f : 0.0f);
      const float xi = clamp_nanprop((fl[j] + r) + p.zp, p.lo, p.hi);      // zp = 0 on a symmetric grid
      o[j] = q_dequant(xi, p);
    }
    reinterpret_cast<vec*>(out)[iv] = o;
  }
}

__global__ __launch_bounds__(kBlock) void ada_init_k(const float* __restrict__ w, float* __restrict__ alpha, uint64_t n,
                                                     tq_quantizer q, int mode, float temp) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const QP p = make_qp<true>(q, par_index(q, i));
    const float x = w[i] / p.scale;
    const float rest = x - floorf(x);
    float a;
    if (mode == TQ_ADA_HARD_SIGMOID) {
      a = -logf((kZeta - rest) / (rest - kGamma));                 // hard_logit, quantizer.py:32-34
    } else {
      const float pr = clamp_nanprop(rest, 1e-16f, (float)(1.0 - 1e-16));
      a = -logf(1.0f / pr - 1.0f);                                 // logit, quantizer.py:22-24
      if (mode == TQ_ADA_SIGMOID_TEMP) a = temp * a;
    }
    alpha[i] = a;
  }
}

__global__ __launch_bounds__(kBlock) void ada_bwd_k(const float* __restrict__ w, const float* __restrict__ alpha,
                                                    const float* __restrict__ g_wq, float* __restrict__ g_alpha,
                                                    uint64_t n, tq_quantizer q, int mode, float temp) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const QP p = make_qp<true>(q, par_index(q, i));
    float dh;
    const float h = ada_h_fast(alpha[i], mode, mode == TQ_ADA_SIGMOID_TEMP ? 1.0f / temp : 1.0f, dh);
    float xi = floor_quot(w[i], p.scale, guarded_rcp(p.scale)) + h;
    if (!q.symmetric) xi += p.zp;
    const bool in = (xi >= p.lo) && (xi <= p.hi);
    g_alpha[i] = in ? (g_wq[i] * p.scale) * dh : 0.0f;
  }
}

template <int V, int MODE>
__global__ __launch_bounds__(kBlock) void ada_bwd_adam_k(const float* __restrict__ w, const float* __restrict__ g_wq,
                                                         float* __restrict__ alpha, float* __restrict__ m,
                                                         float* __restrict__ v, float* __restrict__ g_out, uint64_t n,
                                                         tq_quantizer q, float temp, float reg_w, float beta,
                                                         float lr, float b1, float b2, float adam_eps, float bc1,
                                                         float bc2_sqrt, const float* __restrict__ sched) {
  typedef typename AdaVec<V>::type vec;
  if (sched != nullptr) {      // per-iteration scalars from device memory (hipGraph replay of the optimisation loop)
    reg_w = sched[0]; beta = sched[1]; bc1 = sched[2]; bc2_sqrt = sched[3];
  }
  const uint64_t nv = n / V;
  // uniform scalars of the step, hoisted: every division below is on SGPR values, once per wave
  const float inv_temp = MODE == TQ_ADA_SIGMOID_TEMP ? 1.0f / temp : 1.0f;
  const float one_m_b1 = 1.0f - b1, one_m_b2 = 1.0f - b2;
  const float inv_bc2_sqrt = 1.0f / bc2_sqrt, step_size = lr / bc1;
  const bool reg_on = reg_w != 0.0f;
  const float reg_k = reg_on ? -2.0f * reg_w * beta : 0.0f, bm1 = reg_on ? beta - 1.0f : 1.0f;
  const bool one = q.n_params == 1;
  QP p = make_qp<true>(q, 0);                              // per-tensor grid: loop-invariant
  float rcp = guarded_rcp(p.scale);
  for (uint64_t iv = (uint64_t)blockIdx.x * kBlock + threadIdx.x; iv < nv; iv += (uint64_t)gridDim.x * kBlock) {
    if (!one) {
      p = make_qp<true>(q, par_index(q, iv * V));
      rcp = guarded_rcp(p.scale);
    }
    const vec wv = reinterpret_cast<const vec*>(w)[iv], gv = reinterpret_cast<const vec*>(g_wq)[iv];
    const vec av = reinterpret_cast<const vec*>(alpha)[iv], mv = reinterpret_cast<const vec*>(m)[iv];
    const vec vv = reinterpret_cast<const vec*>(v)[iv];
    float fl[V];
    floor_quot_n<V>(wv, p.scale, rcp, fl);
    vec ao, mo, vo, go;
    // branch-free element body: the V chains (7 quarter-rate transcendentals each) interleave
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float a = av[j];
      float dh;
      const float h = ada_h_fast<MODE>(a, inv_temp, dh);
      const float xi = (fl[j] + h) + p.zp;                             // zp = 0 on a symmetric grid
      const bool in = (xi >= p.lo) && (xi <= p.hi);
      // d loss / d alpha through w_q = s * (x_int - zp)
      float g = in ? (gv[j] * p.scale) * dh : 0.0f;
      // d/dalpha [ reg_w * (1 - (2|h - 0.5|)^beta) ] = -2 reg_w beta u^(beta-1) sgn(h - 0.5) dh     (reg_k = 0 when off)
      const float c = h - 0.5f;
      const float u = fabsf(c) * 2.0f;
      float pw = fast_pow(u, bm1);                                     // u = 0: log2 = -inf -> 0 for beta > 1
      pw = (u == 0.0f) ? 0.0f : pw;                                    // (beta == 1: 0 * -inf; the derivative at the kink is 0)
      g = __builtin_fmaf(reg_k * __builtin_copysignf(pw, c), dh, g);
      go[j] = g;
      // torch.optim.Adam (no weight decay / amsgrad)
      const float mi = __builtin_fmaf(g - mv[j], one_m_b1, mv[j]);     // exp_avg.lerp_(grad, 1 - beta1)
      const float vi = __builtin_fmaf(one_m_b2 * g, g, vv[j] * b2);    // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
      const float denom = __builtin_fmaf(__builtin_amdgcn_sqrtf(vi), inv_bc2_sqrt, adam_eps);
      ao[j] = __builtin_fmaf(-step_size * mi, __builtin_amdgcn_rcpf(denom), a);
      mo[j] = mi;
      vo[j] = vi;
    }
    if (g_out) reinterpret_cast<vec*>(g_out)[iv] = go;
    reinterpret_cast<vec*>(alpha)[iv] = ao;
    reinterpret_cast<vec*>(m)[iv] = mo;
    reinterpret_cast<vec*>(v)[iv] = vo;
  }
}

// 16-byte vectors when every stream allows it
static bool ada_vec4(uint64_t n, const tq_quantizer* q, std::initializer_list<const void*> ptrs) {
  if (n % 4 != 0 || (q->n_params != 1 && q->inner % 4 != 0)) return false;
  for (const void* p : ptrs)
    if (p != nullptr && !aligned16(p)) return false;
  return true;
}

// block partial sums -> ws[blockIdx.x]; finalize adds/sets out[0]
__global__ __launch_bounds__(kBlock) void ada_reg_k(const float* __restrict__ alpha, uint64_t n, int mode, float temp,
                                                    float beta, double* __restrict__ ws) {
  __shared__ double s_w[kBlock / kWave];
  double acc = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const float h = ada_h(alpha[i], mode, temp, nullptr);
    acc += (double)(1.0f - powf(fabsf(h - 0.5f) * 2.0f, beta));
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & (kWave - 1)) == 0) s_w[threadIdx.x / kWave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < kBlock / kWave; ++k) t += s_w[k];
    ws[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(kBlock) void sqdiff_k(const float* __restrict__ a, const float* __restrict__ b, uint64_t n,
                                                   double* __restrict__ ws) {
  __shared__ double s_w[kBlock / kWave];
  double acc = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const float d = a[i] - b[i];
    acc += (double)(d * d);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & (kWave - 1)) == 0) s_w[threadIdx.x / kWave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < kBlock / kWave; ++k) t += s_w[k];
    ws[blockIdx.x] = t;
  }
}

__global__ void reduce_final_k(const double* __restrict__ ws, uint32_t nb, double scale, int accumulate,
                               double* __restrict__ out) {
  __shared__ double s_w[kBlock / kWave];
  double acc = 0.0;
  for (uint32_t i = threadIdx.x; i < nb; i += kBlock) acc += ws[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & (kWave - 1)) == 0) s_w[threadIdx.x / kWave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < kBlock / kWave; ++k) t += s_w[k];
    out[0] = accumulate ? out[0] + t * scale : t * scale;
  }
}

static unsigned ew_grid(uint64_t n) {
  return (unsigned)std::min<uint64_t>(std::max<uint64_t>(ceil_div(n, kBlock), 1), kMaxGrid);
}

static int check_mode(int mode, float temp, const char* who) {
  if (mode < TQ_ADA_SIGMOID || mode > TQ_ADA_SIGMOID_TEMP) return set_error(TQ_EINVAL, "%s: unknown round mode %d", who, mode);
  if (mode == TQ_ADA_SIGMOID_TEMP && !(temp > 0.0f)) return set_error(TQ_EINVAL, "%s: temperature must be > 0", who);
  return TQ_OK;
}

}  // namespace tq

using namespace tq;

// dispatch of the templated per-iteration kernels over (vector width, relaxation)
#define TQ_ADA_MODES(LAUNCH)                                               \
  switch (mode) {                                                          \
    case TQ_ADA_SIGMOID: LAUNCH(TQ_ADA_SIGMOID); break;                    \
    case TQ_ADA_HARD_SIGMOID: LAUNCH(TQ_ADA_HARD_SIGMOID); break;          \
    default: LAUNCH(TQ_ADA_SIGMOID_TEMP); break;                           \
  }

extern "C" int tq_adaround_fwd(const float* w, const float* alpha, float* w_q, uint64_t n, const tq_quantizer* q,
                               int mode, int soft, float temperature, tq_stream_t stream) {
  TQ_REQUIRE(w && alpha && w_q, "tq_adaround_fwd: NULL pointer");
  if (int e = check_quantizer(q, n, "tq_adaround_fwd")) return e;
  if (int e = check_mode(mode, temperature, "tq_adaround_fwd")) return e;
  if (n == 0) return TQ_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool v4 = ada_vec4(n, q, {w, alpha, w_q});
  // one-shot tiles like the step kernel (a 2048-block grid-stride loop left [3072,768] = 2304 tiles with a second round
  // for 256 blocks)
  const unsigned grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>(ceil_div(v4 ? n / 4 : n, kBlock), 1), 1u << 30);
#define TQ_ADA_FWD(M)                                                                                                  \
  do {                                                                                                                 \
    if (v4 && soft) hipLaunchKernelGGL((ada_fwd_k<4, M, true>), dim3(grid), dim3(kBlock), 0, st, w, alpha, w_q, n, *q, temperature);       \
    else if (v4) hipLaunchKernelGGL((ada_fwd_k<4, TQ_ADA_SIGMOID, false>), dim3(grid), dim3(kBlock), 0, st, w, alpha, w_q, n, *q, temperature); \
    else if (soft) hipLaunchKernelGGL((ada_fwd_k<1, M, true>), dim3(grid), dim3(kBlock), 0, st, w, alpha, w_q, n, *q, temperature);        \
    else hipLaunchKernelGGL((ada_fwd_k<1, TQ_ADA_SIGMOID, false>), dim3(grid), dim3(kBlock), 0, st, w, alpha, w_q, n, *q, temperature);    \
  } while (0)
  TQ_ADA_MODES(TQ_ADA_FWD)
#undef TQ_ADA_FWD
  return check_launch("ada_fwd_k");
}

extern "C" int tq_adaround_init_alpha(const float* w, float* alpha, uint64_t n, const tq_quantizer* q, int mode,
                                      float temperature, tq_stream_t stream) {
  TQ_REQUIRE(w && alpha, "tq_adaround_init_alpha: NULL pointer");
  if (int e = check_quantizer(q, n, "tq_adaround_init_alpha")) return e;
  if (int e = check_mode(mode, temperature, "tq_adaround_init_alpha")) return e;
  if (n == 0) return TQ_OK;
  hipLaunchKernelGGL(ada_init_k, dim3(ew_grid(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), w, alpha, n, *q, mode,
                     temperature);
  return check_launch("ada_init_k");
}

extern "C" int tq_adaround_bwd(const float* w, const float* alpha, const float* grad_wq, float* grad_alpha, uint64_t n,
                               const tq_quantizer* q, int mode, float temperature, tq_stream_t stream) {
  TQ_REQUIRE(w && alpha && grad_wq && grad_alpha, "tq_adaround_bwd: NULL pointer");
  if (int e = check_quantizer(q, n, "tq_adaround_bwd")) return e;
  if (int e = check_mode(mode, temperature, "tq_adaround_bwd")) return e;
  if (n == 0) return TQ_OK;
  hipLaunchKernelGGL(ada_bwd_k, dim3(ew_grid(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), w, alpha, grad_wq,
                     grad_alpha, n, *q, mode, temperature);
  return check_launch("ada_bwd_k");
}

// one block per 256 vectors (one-shot tiling, like K1): the capped grid-stride form lost ~10 % on the 656 MB embedding step
static unsigned ada_grid(uint64_t work) {
  const int cap = tuning("TQ_ADA_GRID", 0);           // 0 = one-shot; > 0 caps the grid (grid-stride loop): A/B only
  return (unsigned)std::min<uint64_t>(std::max<uint64_t>(ceil_div(work, kBlock), 1), cap > 0 ? (uint64_t)cap : (1u << 30));
}

static void launch_bwd_adam(bool v4, int mode, hipStream_t st, const float* w, const float* grad_wq, float* alpha,
                            float* exp_avg, float* exp_avg_sq, float* g_out, uint64_t n, const tq_quantizer* q, float temperature,
                            float reg_weight, float beta, float lr, float b1, float b2, float adam_eps, float bc1, float bc2_sqrt,
                            const float* sched) {
  const unsigned grid = ada_grid(v4 ? n / 4 : n);
#define TQ_ADA_STEP(M)                                                                                                     \
  do {                                                                                                                     \
    if (v4) hipLaunchKernelGGL((ada_bwd_adam_k<4, M>), dim3(grid), dim3(kBlock), 0, st, w, grad_wq, alpha, exp_avg, exp_avg_sq, g_out, n, \
                               *q, temperature, reg_weight, beta, lr, b1, b2, adam_eps, bc1, bc2_sqrt, sched);            \
    else hipLaunchKernelGGL((ada_bwd_adam_k<1, M>), dim3(grid), dim3(kBlock), 0, st, w, grad_wq, alpha, exp_avg, exp_avg_sq, g_out, n,   \
                            *q, temperature, reg_weight, beta, lr, b1, b2, adam_eps, bc1, bc2_sqrt, sched);               \
  } while (0)
  TQ_ADA_MODES(TQ_ADA_STEP)
#undef TQ_ADA_STEP
}

extern "C" int tq_adaround_bwd_adam(const float* w, const float* grad_wq, float* alpha, float* exp_avg,
                                    float* exp_avg_sq, float* grad_alpha_out, uint64_t n, const tq_quantizer* q,
                                    int mode, float temperature, float reg_weight, float beta, float lr,
                                    float adam_b1, float adam_b2, float adam_eps, int step, tq_stream_t stream) {
  TQ_REQUIRE(w && grad_wq && alpha && exp_avg && exp_avg_sq, "tq_adaround_bwd_adam: NULL pointer");
  TQ_REQUIRE(step >= 1, "tq_adaround_bwd_adam: step must be >= 1");
  if (int e = check_quantizer(q, n, "tq_adaround_bwd_adam")) return e;
  if (int e = check_mode(mode, temperature, "tq_adaround_bwd_adam")) return e;
  if (n == 0) return TQ_OK;
  // bias corrections in double like python, narrowed once (torch computes them as python floats)
  const double bc1 = 1.0 - pow((double)adam_b1, (double)step);
  const double bc2 = 1.0 - pow((double)adam_b2, (double)step);
  launch_bwd_adam(ada_vec4(n, q, {w, grad_wq, alpha, exp_avg, exp_avg_sq, grad_alpha_out}), mode,
                  static_cast<hipStream_t>(stream), w, grad_wq, alpha, exp_avg, exp_avg_sq, grad_alpha_out, n, q, temperature,
                  reg_weight, beta, lr, adam_b1, adam_b2, adam_eps, (float)bc1, (float)sqrt(bc2), nullptr);
  return check_launch("ada_bwd_adam_k");
}

// The same step with its four per-iteration scalars read from DEVICE memory: sched = {reg_weight, beta,
// 1 - b1^t, sqrt(1 - b2^t)} (fp32, prepared by the host for every iteration up front).  No launch argument changes
// from one iteration to the next, so the whole AdaRound iteration can be recorded once as a hipGraph and replayed.
extern "C" int tq_adaround_bwd_adam_sched(const float* w, const float* grad_wq, float* alpha, float* exp_avg,
                                          float* exp_avg_sq, uint64_t n, const tq_quantizer* q, int mode, float temperature,
                                          const float* sched, float lr, float adam_b1, float adam_b2, float adam_eps,
                                          tq_stream_t stream) {
  TQ_REQUIRE(w && grad_wq && alpha && exp_avg && exp_avg_sq && sched, "tq_adaround_bwd_adam_sched: NULL pointer");
  if (int e = check_quantizer(q, n, "tq_adaround_bwd_adam_sched")) return e;
  if (int e = check_mode(mode, temperature, "tq_adaround_bwd_adam_sched")) return e;
  if (n == 0) return TQ_OK;
  launch_bwd_adam(ada_vec4(n, q, {w, grad_wq, alpha, exp_avg, exp_avg_sq}), mode, static_cast<hipStream_t>(stream), w, grad_wq,
                  alpha, exp_avg, exp_avg_sq, nullptr, n, q, temperature, 0.0f, 0.0f, lr, adam_b1, adam_b2, adam_eps, 1.0f, 1.0f,
                  sched);
  return check_launch("ada_bwd_adam_k");
}

extern "C" size_t tq_reduce_workspace_bytes(uint64_t n) { return (size_t)kMaxGrid * sizeof(double); }

extern "C" int tq_adaround_reg(const float* alpha, uint64_t n, int mode, float temperature, float beta, float weight,
                               double* out, void* workspace, size_t workspace_bytes, tq_stream_t stream) {
  TQ_REQUIRE(alpha && out, "tq_adaround_reg: NULL pointer");
  TQ_REQUIRE(workspace && workspace_bytes >= (size_t)kMaxGrid * sizeof(double), "tq_adaround_reg: workspace too small");
  if (int e = check_mode(mode, temperature, "tq_adaround_reg")) return e;
  if (n == 0) return TQ_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned g = ew_grid(n);
  hipLaunchKernelGGL(ada_reg_k, dim3(g), dim3(kBlock), 0, st, alpha, n, mode, temperature, beta, static_cast<double*>(workspace));
  hipLaunchKernelGGL(reduce_final_k, dim3(1), dim3(kBlock), 0, st, static_cast<const double*>(workspace), g, (double)weight, 1, out);
  return check_launch("tq_adaround_reg");
}

extern "C" int tq_recon_loss(const float* pred, const float* tgt, uint64_t d0, uint64_t d1, uint64_t rest, double* out,
                             void* workspace, size_t workspace_bytes, tq_stream_t stream) {
  TQ_REQUIRE(pred && tgt && out, "tq_recon_loss: NULL pointer");
  TQ_REQUIRE(d0 >= 1 && d1 >= 1 && rest >= 1, "tq_recon_loss: empty tensor");
  TQ_REQUIRE(workspace && workspace_bytes >= (size_t)kMaxGrid * sizeof(double), "tq_recon_loss: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const uint64_t n = d0 * d1 * rest;
  const unsigned g = ew_grid(n);
  hipLaunchKernelGGL(sqdiff_k, dim3(g), dim3(kBlock), 0, st, pred, tgt, n, static_cast<double*>(workspace));
  // sum over dim 1, mean over the remaining d0 * rest positions
  hipLaunchKernelGGL(reduce_final_k, dim3(1), dim3(kBlock), 0, st, static_cast<const double*>(workspace), g,
                     1.0 / (double)(d0 * rest), 0, out);
  return check_launch("tq_recon_loss");
}
