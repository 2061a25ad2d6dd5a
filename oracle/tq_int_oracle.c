/* CPU oracle, plain C: the INTEGER evaluation of fixed-range quantized layers.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/tq_oracle.py): built by oracle/Makefile into oracle/_build/libtq_oracle.so and
 * called from tests/ (and tests/_oracle_backend.py) as the checker of tq_linear_i8_fwd / tq_linear_i8_nonorm_fwd /
 * tq_ffn_i8_nonorm_fwd / tq_attention_i8_fwd.  Compile WITHOUT -ffast-math and with -ffp-contract=off: every float
 * operation below is ONE IEEE fp32 operation (fmaf where the kernels use a fused multiply-add), so the results are
 * comparable BIT FOR BIT with the kernels -- there is no fp32 GEMM anywhere: the contractions are exact integers.
 *
 * What is restated (reference files relative to the upstream repository root):
 *   quantization/hijacker.py:66-116 + autoquant_utils.py:16-21   y = Q_out(act(F.linear(Q_x(x), Q_w(W), b)))
 *       with Q_x(x) = s_x (a - z_x), Q_w(W) = s_w w:  sum_k Q_x Q_w = s_x s_w (sum_k a w - z_x sum_k w)   -- exact in
 *       integers; the reference's fp32 GEMM evaluates the same number up to its own accumulation round-off.
 *   quantization/quantizers.py:132-153,184-185,209             the quantizer Q (asymmetric / symmetric)
 *   models/quantized_mobilebert.py:58-72, 287-304, 330-352      NoNorm tail: Q(Q(Q(lin) + res) * w + b)
 *   models/quantized_bert.py:135-213                            attention: Q K^T, score quantizer, / sqrt(d), + mask,
 *                                                               softmax, probability quantizer, P V, context quantizer
 * Index operands are int8(index - 128) for activations (asymmetric, <= 8 bit) and int8(index) for symmetric signed
 * weights, exactly what the kernels consume (include/tq_hip.h, tq_linear_i8_fwd).
 *
 * Two places are specified BY THIS FILE rather than by the reference (whose torch.exp / torch.erf are not reproducible
 * bit for bit on any other implementation): the softmax exponential `tq_exp_neg` (Cephes-style range reduction +
 * degree-5 polynomial, IEEE operations only -- csrc/tq_device.h holds the identical device function) and GELU, in two
 * forms: activation code 4 = the correctly rounded fp32 value of x/2 (1 + erf(x / sqrt 2)) (float64 evaluation, narrowed
 * once), which the kernels reproduce EXACTLY through the staircase table of csrc/tq_stair.hip (compared at zero
 * tolerance), and code 2 = the single minimax fit of the arithmetic epilogue of csrc/tq_linear_i8.hip (its exp2 is the
 * hardware v_exp_f32 on the GPU and exp2f here: results may differ in the last bit, compared at a tolerance).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  float delta, zero_float, eps;
  int32_t n_bits, symmetric, is_signed, present;
} tq_io_q;

typedef struct { float scale, zp, lo, hi; int on; } qp_t;

static float clamp_nanprop(float v, float lo, float hi) {
  if (v != v) return v;
  v = v < lo ? lo : v;
  v = v > hi ? hi : v;
  return v;
}

static qp_t make_qp(const tq_io_q* q) {
  qp_t r = {1.0f, 0.0f, 0.0f, 0.0f, 0};
  if (q == NULL || !q->present) return r;
  r.on = 1;
  r.scale = q->delta < q->eps ? q->eps : q->delta;
  if (q->symmetric) {
    r.zp = 0.0f;
    r.lo = q->is_signed ? -(float)ldexp(1.0, q->n_bits - 1) : 0.0f;
    r.hi = (float)(ldexp(1.0, q->n_bits - (q->is_signed ? 1 : 0)) - 1.0);
  } else {
    r.lo = 0.0f;
    r.hi = (float)(ldexp(1.0, q->n_bits) - 1.0);
    r.zp = clamp_nanprop(rintf(q->zero_float), r.lo, r.hi);
  }
  return r;
}

/* x_int = clamp(rne(x / scale) + zp, lo, hi); y = scale * (x_int - zp) */
static float q_index(float v, const qp_t* p) { return clamp_nanprop(rintf(v / p->scale) + p->zp, p->lo, p->hi); }
static float q_dequant(float xi, const qp_t* p) { return p->scale * (xi - p->zp); }
static float fq(float v, const qp_t* p, float* idx) {
  if (!p->on) return v;
  const float xi = q_index(v, p);
  if (idx) *idx = xi;
  return q_dequant(xi, p);
}
static int8_t idx_m128(float xi) { return (int8_t)((int)xi - 128); }

/* exp(x) for x <= 0 from IEEE operations only; x < -86 (and -inf) -> 0, NaN -> NaN.  Identical to exp_neg_ieee in
 * csrc/tq_device.h.  |relative error| < 2 ulp on [-86, 0].                                                        */
float tq_exp_neg(float x) {
  if (x != x) return x;
  if (x < -86.0f) return 0.0f;
  const float k = rintf(x * 1.44269504088896341f);
  float r = fmaf(k, -0.693359375f, x);
  r = fmaf(k, 2.12194440e-4f, r);
  const float z = r * r;
  float y = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
  y = fmaf(y, r, 8.3334519073e-3f);
  y = fmaf(y, r, 4.1665795894e-2f);
  y = fmaf(y, r, 1.6666665459e-1f);
  y = fmaf(y, r, 5.0000001201e-1f);
  y = fmaf(y, z, r);
  y = y + 1.0f;
  return ldexpf(y, (int)k);
}

/* GELU(v) = 0.5 v (1 + erf(v / sqrt 2)) with the single fit of csrc/tq_linear_i8.hip (gelu_erf_n, round 4):
 * erfc(t) = 2^(-t Q(t)), t = |v| / sqrt 2, Q of degree 7 -- the same single IEEE operations in the same order; exp2f here,
 * v_exp_f32 on the device (<= 1 ulp apart: the one documented exception of the integer path's bit-exactness) */
static float gelu_fit(float v) {
  const float a = v * 0.70710678118654752440f;
  const float t = fabsf(a);
  float q = fmaf(t, -4.5358574425335974e-05f, 0.00044550743768922985f);
  q = fmaf(t, q, -0.0014894409105181694f);
  q = fmaf(t, q, -0.0007746309274807572f);
  q = fmaf(t, q, 0.02825368195772171f);
  q = fmaf(t, q, -0.14848162233829498f);
  q = fmaf(t, q, -0.9184163808822632f);
  q = fmaf(t, q, -1.6279085874557495f);
  q = q * t;
  const float e = 1.0f - exp2f(q);
  const float r = copysignf(e, a);
  return (v * 0.5f) * (1.0f + r);
}

/* act 4: the specification of the staircase epilogue (csrc/tq_stair.hip, tq_linear_i8_stair_fwd with a table whose
 * header says ok): the CORRECTLY ROUNDED fp32 value of nn.GELU()'s erf form -- float64 evaluation, narrowed once -- in
 * front of the reference's own fp32 quantizer arithmetic.  No table here: every element is evaluated directly. */
static float gelu_rn32(float v) {
  const double x = (double)v;
  return (float)(0.5 * x * (1.0 + erf(x * 0.70710678118654752440)));
}

static float act_fn(float v, int act) {
  switch (act) {
    case 1: return v > 0.0f ? v : 0.0f;
    case 2: return gelu_fit(v);
    case 3: return tanhf(v);
    case 4: return gelu_rn32(v);
    default: return v;
  }
}

static int zero_point_i(float zero_float, int n_bits) {
  return (int)clamp_nanprop(rintf(zero_float), 0.0f, (float)(ldexp(1.0, n_bits) - 1.0));
}

/* pre-activation of the integer Linear: s_x s_w[n] (sum_k a'_x w + (128 - z_x) rowsum[n]) + b[n], a' = index - 128 */
static float lin_pre(const int8_t* x, const int8_t* w, int64_t K, int shift, float sx, float dw, float w_eps, float b) {
  int32_t acc = 0, rs = 0;                                    /* exact: |x w| <= 2^14, K <= 16384 (the kernels use i32 too) */
  for (int64_t k = 0; k < K; ++k) {
    acc += (int32_t)x[k] * (int32_t)w[k];
    rs += w[k];
  }
  const int32_t tot = acc + rs * shift;
  const float sw = sx * (dw < w_eps ? w_eps : dw);
  return (float)tot * sw + b;
}

/* tail: 0 none | 1: Q_t2(Q_out(v) * nn_w + nn_b) | 2: Q_t2(Q_t1(Q_out(v) + residual) * nn_w + nn_b).
 * y_idx (optional): int8(index - 128) on the grid of the LAST active quantizer of the chain (Q_t2 with a tail).   */
void tq_io_linear_i8(const int8_t* x_idx, const int8_t* w_idx, const float* bias, float* y, int8_t* y_idx, int64_t M,
                     int64_t N, int64_t K, float x_delta, float x_zero_float, int x_n_bits, float x_eps,
                     const float* w_delta, int64_t w_n_params, float w_eps, int act, const tq_io_q* q_out, int tail,
                     const float* residual, const float* nn_w, const float* nn_b, const tq_io_q* q_t1,
                     const tq_io_q* q_t2) {
  const float sx = x_delta < x_eps ? x_eps : x_delta;
  const int shift = 128 - zero_point_i(x_zero_float, x_n_bits);
  const qp_t qo = make_qp(q_out), q1 = make_qp(q_t1), q2 = make_qp(q_t2);
  for (int64_t m = 0; m < M; ++m)
    for (int64_t n = 0; n < N; ++n) {
      float v = lin_pre(x_idx + m * K, w_idx + n * K, K, shift, sx, w_delta[w_n_params == 1 ? 0 : n], w_eps,
                        bias ? bias[n] : 0.0f);
      v = act_fn(v, act);
      float xi = 0.0f;
      v = fq(v, &qo, &xi);
      if (tail == 2) {
        v = v + residual[m * N + n];
        v = fq(v, &q1, NULL);
      }
      if (tail >= 1) {
        v = v * nn_w[n] + nn_b[n];
        v = fq(v, &q2, &xi);
      }
      y[m * N + n] = v;
      if (y_idx) y_idx[m * N + n] = idx_m128(xi);
    }
}

/* MobileBERT feed-forward block: lin2(Q_mid(relu(lin1(x)))) with the residual NoNorm tail; the intermediate lives on
 * q_mid's grid as int8(index - 128), exactly what tq_ffn_i8_nonorm_fwd keeps in LDS.                               */
void tq_io_ffn_i8(const int8_t* x_idx, float x_delta, float x_zero_float, int x_n_bits, float x_eps, const int8_t* w1_idx,
                  const float* bias1, const float* w1_delta, int64_t w1_n_params, float w1_eps, const tq_io_q* q_mid,
                  const int8_t* w2_idx, const float* bias2, const float* w2_delta, int64_t w2_n_params, float w2_eps,
                  const float* residual, const float* nn_w, const float* nn_b, const tq_io_q* q_dense,
                  const tq_io_q* q_sum, const tq_io_q* q_out, float* y, int8_t* y_idx, int64_t M, int64_t K1, int64_t N1,
                  int64_t N2) {
  float* mid = (float*)malloc((size_t)M * N1 * sizeof(float));
  int8_t* mid_idx = (int8_t*)malloc((size_t)M * N1);
  tq_io_linear_i8(x_idx, w1_idx, bias1, mid, mid_idx, M, N1, K1, x_delta, x_zero_float, x_n_bits, x_eps, w1_delta, w1_n_params,
                  w1_eps, 1 /* ReLU */, q_mid, 0, NULL, NULL, NULL, NULL, NULL);
  tq_io_linear_i8(mid_idx, w2_idx, bias2, y, y_idx, M, N2, N1, q_mid->delta, q_mid->zero_float, q_mid->n_bits, q_mid->eps,
                  w2_delta, w2_n_params, w2_eps, 0, q_dense, 2, residual, nn_w, nn_b, q_sum, q_out);
  free(mid);
  free(mid_idx);
}

/* Attention core.  q / k / v: int8(index - 128) [B, T, H * D] with token stride `stride` (H * D, or 3 H D inside a
 * stacked Q|K|V buffer).  Summation order of the softmax denominator (it decides the last bit of `sum`), as the kernel
 * defines it: keys are split in two halves of T / 2; inside a half, key 16 t + 4 g + r belongs to lane group g
 * (g = 0..3); a group adds its exponentials sequentially over t (outer) and r (inner); the half's sum is
 * (s_0 + s_1) + (s_2 + s_3); the row's sum is half_0 + half_1.                                                  */
void tq_io_attention_i8(const int8_t* q, const int8_t* k, const int8_t* v, float* ctx, int8_t* ctx_idx, int64_t B, int64_t T,
                        int64_t H, int64_t D, int64_t stride, const float* mask, float denom, const tq_io_q* q_q,
                        const tq_io_q* q_k, const tq_io_q* q_v, const tq_io_q* q_scores, const tq_io_q* q_probs,
                        const tq_io_q* q_ctx) {
  const qp_t pq = make_qp(q_q), pk = make_qp(q_k), pv = make_qp(q_v), pp = make_qp(q_probs);
  const qp_t ps = make_qp(q_scores), pc = make_qp(q_ctx);
  const int cq = 128 - (int)pq.zp, ck = 128 - (int)pk.zp, cv = 128 - (int)pv.zp, cp = 128 - (int)pp.zp;
  const float s_qk = pq.scale * pk.scale, s_pv = pp.scale * pv.scale;
  float* e = (float*)malloc((size_t)T * sizeof(float));
  int* pidx = (int*)malloc((size_t)T * sizeof(int));
  const int64_t half = T / 2, tiles = half / 16;
  for (int64_t b = 0; b < B; ++b)
    for (int64_t h = 0; h < H; ++h)
      for (int64_t i = 0; i < T; ++i) {
        const int8_t* qi = q + (b * T + i) * stride + h * D;
        int rsq = 0;
        for (int64_t d = 0; d < D; ++d) rsq += qi[d];
        float mx = -INFINITY;
        for (int64_t j = 0; j < T; ++j) {
          const int8_t* kj = k + (b * T + j) * stride + h * D;
          int acc = 0, rsk = 0;
          for (int64_t d = 0; d < D; ++d) { acc += (int)qi[d] * (int)kj[d]; rsk += kj[d]; }
          float x = (float)(acc + cq * rsk + ck * rsq + (int)D * cq * ck) * s_qk;
          x = fq(x, &ps, NULL);
          x = x / denom;
          if (mask) x = x + mask[b * T + j];
          e[j] = x;
          mx = fmaxf(mx, x);
        }
        int any_nan = 0;
        for (int64_t j = 0; j < T; ++j) any_nan |= (e[j] != e[j]);
        float sum;
        {
          float hs[2];
          for (int kh = 0; kh < 2; ++kh) {
            float sg[4];
            for (int g = 0; g < 4; ++g) {
              float s = 0.0f;
              for (int64_t t = 0; t < tiles; ++t)
                for (int r = 0; r < 4; ++r) {
                  const int64_t j = kh * half + 16 * t + 4 * g + r;
                  const float ex = tq_exp_neg(e[j] - mx);
                  e[j] = ex;
                  s += ex;
                }
              sg[g] = s;
            }
            hs[kh] = (sg[0] + sg[1]) + (sg[2] + sg[3]);
          }
          sum = hs[0] + hs[1];
        }
        const int bad_row = (sum != sum) || any_nan;
        int rsp = 0;
        for (int64_t j = 0; j < T; ++j) {
          const float pr = e[j] / sum;
          const float xi = clamp_nanprop(rintf(pr / pp.scale) + pp.zp, pp.lo, pp.hi);
          pidx[j] = (xi != xi) ? 0 : (int)xi - 128;
          rsp += pidx[j];
        }
        for (int64_t d = 0; d < D; ++d) {
          int acc = 0, csv = 0;
          for (int64_t j = 0; j < T; ++j) {
            const int a = v[(b * T + j) * stride + h * D + d];
            acc += pidx[j] * a;
            csv += a;
          }
          float c = (float)(acc + cp * csv + cv * rsp + (int)T * cp * cv) * s_pv;
          float xi = 0.0f;
          c = fq(c, &pc, &xi);
          const size_t o = (size_t)((b * T + i) * H * D + h * D + d);
          ctx[o] = bad_row ? NAN : c;
          if (ctx_idx) ctx_idx[o] = idx_m128(xi);
        }
      }
  free(e);
  free(pidx);
}
