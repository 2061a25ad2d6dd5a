/* CPU oracle, plain C: an independent restatement of the core fake-quant arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/tq_oracle.py).  Built by oracle/Makefile into
 * oracle/_build/libtq_oracle.so; loaded by tests/ and __graft_entry__.smoke() as a second,
 * torch-free checker of rounding / clamping / division semantics.  Must be compiled WITHOUT
 * -ffast-math and with -ffp-contract=off so that every operation is a single IEEE fp32 op.
 *
 * Follows the reference's quantization/quantizers.py:
 *   scale = max(delta, eps)                                   :142-145
 *   zp    = clamp(rne(zero_float), 0, 2^n - 1)  (asym) | 0    :149-153, :330-332
 *   x_int = clamp(rne(x / scale) + zp, lo, hi)                :184-185
 *   y     = scale * (x_int - zp)                              :209
 *   range -> (delta, zero_float)                              :258-259, :276-277, :335-339
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

static float clampf_(float v, float lo, float hi) {
  v = v < lo ? lo : v;
  v = v > hi ? hi : v;
  return v;
}

/* params: p = (i / inner) % n_params; zero_float == NULL -> symmetric with `is_signed` */
void tq_oracle_fake_quant_f32(const float* x, float* y, float* idx, size_t n, const float* delta,
                              const float* zero_float, int is_signed, int n_bits, float eps,
                              size_t n_params, size_t inner) {
  for (size_t i = 0; i < n; ++i) {
    const size_t p = n_params == 1 ? 0 : (i / inner) % n_params;
    const float scale = delta[p] < eps ? eps : delta[p];
    float lo, hi, zp;
    if (zero_float) {
      lo = 0.0f;
      hi = (float)(ldexp(1.0, n_bits) - 1.0);
      zp = clampf_(rintf(zero_float[p]), lo, hi);
    } else {
      zp = 0.0f;
      lo = is_signed ? -(float)ldexp(1.0, n_bits - 1) : 0.0f;
      hi = (float)(ldexp(1.0, n_bits - (is_signed ? 1 : 0)) - 1.0);
    }
    const float xi = clampf_(rintf(x[i] / scale) + zp, lo, hi);
    if (idx) idx[i] = xi;
    if (y) y[i] = scale * (xi - zp);
  }
}

void tq_oracle_minmax_f32(const float* x, size_t n, float* out_min, float* out_max) {
  float mn = INFINITY, mx = -INFINITY;
  for (size_t i = 0; i < n; ++i) {
    if (x[i] < mn) mn = x[i];
    if (x[i] > mx) mx = x[i];
  }
  *out_min = mn;
  *out_max = mx;
}

void tq_oracle_range_to_asym(float x_min, float x_max, int n_bits, float eps, float* delta,
                             float* zero_float) {
  const float lo = x_min < 0.0f ? x_min : 0.0f;
  const float hi = x_max > eps ? x_max : eps;
  *delta = (hi - lo) / (float)(ldexp(1.0, n_bits) - 1.0);
  *zero_float = (-lo) / *delta;
}

void tq_oracle_range_to_sym(float x_min, float x_max, int n_bits, float eps, float* delta,
                            int* is_signed) {
  const float lo = x_min < 0.0f ? x_min : 0.0f;
  const float hi = x_max > eps ? x_max : eps;
  *is_signed = lo < 0.0f;
  const float a = fabsf(lo) > hi ? fabsf(lo) : hi;
  *delta = a / (float)(ldexp(1.0, n_bits - (*is_signed ? 1 : 0)) - 1.0);
}

/* sum of squared quantization error of one candidate (scale, zp, lo, hi), fp64 accumulate */
double tq_oracle_mse_f32(const float* x, size_t n, float scale, float zp, float lo, float hi) {
  double acc = 0.0;
  for (size_t i = 0; i < n; ++i) {
    const float xi = clampf_(rintf(x[i] / scale) + zp, lo, hi);
    const float d = x[i] - scale * (xi - zp);
    acc += (double)(d * d);
  }
  return acc;
}
