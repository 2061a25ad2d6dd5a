// FP64 variants of the quantizer path (`--double`, reference main.py:227-231: every module with a weight or bias is
// cast to float64, so activations, statistics and range buffers are double).  A numerical-debugging mode in the
// reference and here: same arithmetic as the fp32 kernels with every operation in IEEE double, straightforward
// layouts (16-byte vectors where the tensor allows, HBM-bound at 16 B/element), no exact-quotient tricks -- the
// FP64 division is what the reference executes.  Parameters are double arrays (a quantizer whose range came from python
// floats holds fp32 buffers in the reference, quantizers.py:248-250; torch promotes them in `x / scale`, the host side
// widens them the same way -- exactly -- before the call).
//
//   tq_fake_quant_fwd_f64      quantizers.py:172-211, 291-349       any (n_params, inner) layout
//   tq_fake_quant_bwd_f64      autograd of the same ops (STE round)  gx, optional d_delta / d_zero_float per parameter
//   tq_minmax_f64              range_estimators.py:82-85, 114-130    two-stage min / max with NaN propagation
//   tq_range_update_f64        :87-112, 162-167, 183-193, 209-214    current / all-time / running (+ group fold)
//   tq_axis_ranges_f64         :68-80
//   tq_set_range_{asym,sym}_f64  quantizers.py:234-282, 334-344
//   tq_mse_candidates_f64      range_estimators.py:248-256           candidate table stays fp32 (python-float thresholds
//                              become fp32 tensors in the reference), element arithmetic and sums are double
#include <algorithm>

#include "tq_device.h"
#include "tq_host.h"

namespace tq {

typedef double f64x2 __attribute__((ext_vector_type(2)));

struct QPD {
  double scale, zp, lo, hi;
};

__device__ __forceinline__ double dclamp_nanprop(double v, double lo, double hi) {
  v = v < lo ? lo : v;
  v = v > hi ? hi : v;
  return v;
}
__device__ __forceinline__ double dmin_nanprop(double a, double b) { return (a != a) ? a : ((b != b) ? b : (b < a ? b : a)); }
__device__ __forceinline__ double dmax_nanprop(double a, double b) { return (a != a) ? a : ((b != b) ? b : (b > a ? b : a)); }

__device__ __forceinline__ QPD make_qpd(const tq_quantizer_f64& q, uint64_t p) {
  QPD r;
  const double d = q.delta[p];
  r.scale = q.log_domain ? exp(d) : (d < q.eps ? q.eps : d);       // quantizers.py:142-147
  if (q.symmetric) {
    const bool sgn = q.signed_flag != nullptr && q.signed_flag[0] != 0;
    r.zp = 0.0;
    r.lo = sgn ? -ldexp(1.0, q.n_bits - 1) : 0.0;                    // :321-323
    r.hi = ldexp(1.0, q.n_bits - (sgn ? 1 : 0)) - 1.0;               // :325-328
  } else {
    r.lo = 0.0;
    r.hi = ldexp(1.0, q.n_bits) - 1.0;                               // :137-140
    r.zp = dclamp_nanprop(rint(q.zero_float[p]), r.lo, r.hi);        // :149-153
  }
  return r;
}

__device__ __forceinline__ double qd_index(double x, const QPD& p) { return dclamp_nanprop(rint(x / p.scale) + p.zp, p.lo, p.hi); }
__device__ __forceinline__ double qd_dequant(double xi, const QPD& p) { return p.scale * (xi - p.zp); }

// ------------------------------------------------------------------------------ forward
// VEC: n even and 16-byte aligned pointers -> two elements per thread and access; inner even or per-tensor keeps
// both elements of a pair on the same parameter only when inner % 2 == 0, hence the per-element parameter lookup.
template <bool VEC>
__global__ __launch_bounds__(kBlock) void fq_f64_k(const double* __restrict__ x, double* __restrict__ y, double* __restrict__ idx,
                                                   uint64_t n, tq_quantizer_f64 q) {
  const bool per_tensor = q.n_params == 1;
  QPD p0 = make_qpd(q, 0);
  const uint64_t step = (uint64_t)gridDim.x * kBlock;
  if (VEC) {
    const uint64_t nv = n / 2;
    for (uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x; v < nv; v += step) {
      const f64x2 xv = reinterpret_cast<const f64x2*>(x)[v];
      const QPD pa = per_tensor ? p0 : make_qpd(q, ((2 * v) / q.inner) % q.n_params);
      const QPD pb = per_tensor ? p0 : make_qpd(q, ((2 * v + 1) / q.inner) % q.n_params);
      const double ia = qd_index(xv.x, pa), ib = qd_index(xv.y, pb);
      if (idx != nullptr) reinterpret_cast<f64x2*>(idx)[v] = f64x2{ia, ib};
      if (y != nullptr) reinterpret_cast<f64x2*>(y)[v] = f64x2{qd_dequant(ia, pa), qd_dequant(ib, pb)};
    }
  } else {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += step) {
      const QPD p = per_tensor ? p0 : make_qpd(q, (i / q.inner) % q.n_params);
      const double xi = qd_index(x[i], p);
      if (idx != nullptr) idx[i] = xi;
      if (y != nullptr) y[i] = qd_dequant(xi, p);
    }
  }
}

// ------------------------------------------------------------------------------ STE backward
// dx = ((g * scale) * mask) / scale;  d_delta, d_zero_float through scale = clamp(delta, eps) | exp(delta),
// zp = clamp(round_ste(zero_float), lo, hi)  (same chain as the fp32 kernels, csrc/tq_fake_quant.hip)
__device__ __forceinline__ double ste_bwd_f64(double xv, double g, const QPD& p, bool pgrad, double& acc_d, double& acc_z) {
  const double r = rint(xv / p.scale) + p.zp;
  const bool in = (r >= p.lo) && (r <= p.hi);
  const double gs = g * p.scale;
  if (pgrad) {
    const double xi = dclamp_nanprop(r, p.lo, p.hi);
    double dd = g * (xi - p.zp);
    if (in) dd -= (gs * xv) / (p.scale * p.scale);
    acc_d += dd;
    acc_z += in ? 0.0 : -gs;
  }
  return in ? gs / p.scale : 0.0;
}

__global__ __launch_bounds__(kBlock) void fq_bwd_f64_k(const double* __restrict__ x, const double* __restrict__ gy,
                                                       double* __restrict__ gx, uint64_t n, tq_quantizer_f64 q) {
  const bool per_tensor = q.n_params == 1;
  const QPD p0 = make_qpd(q, 0);
  double d = 0.0, z = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const QPD p = per_tensor ? p0 : make_qpd(q, (i / q.inner) % q.n_params);
    gx[i] = ste_bwd_f64(x[i], gy[i], p, false, d, z);
  }
}

// x viewed as [outer, n_params, inner]; block (p, s) reduces parameter p over slice s of `outer` (deterministic:
// block partials, then a final kernel)
constexpr unsigned kBwdSlicesF64 = 256;
__host__ __device__ inline unsigned bwd_slices_f64(uint64_t outer, uint64_t inner) {
  uint64_t s = outer * inner / 4096;
  if (s < 1) s = 1;
  if (s > kBwdSlicesF64) s = kBwdSlicesF64;
  if (s > outer) s = outer;
  return (unsigned)s;
}

__device__ __forceinline__ void block_sum2(double& a, double& b) {
  __shared__ double s_red[2][kBlock / kWave];
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = threadIdx.x / kWave;
  if ((threadIdx.x & (kWave - 1)) == 0) { s_red[0][w] = a; s_red[1][w] = b; }
  __syncthreads();
  a = b = 0.0;
  for (int k = 0; k < kBlock / kWave; ++k) { a += s_red[0][k]; b += s_red[1][k]; }
  __syncthreads();                                      // s_red may be reused by the caller's next reduction
}

__global__ __launch_bounds__(kBlock) void fq_bwd_params_f64_k(const double* __restrict__ x, const double* __restrict__ gy,
                                                              uint64_t outer, tq_quantizer_f64 q, double* __restrict__ partial) {
  const uint64_t prm = blockIdx.x;
  const QPD p = make_qpd(q, prm);
  const uint64_t chunk = (outer + gridDim.y - 1) / gridDim.y;
  const uint64_t o0 = (uint64_t)blockIdx.y * chunk, o1 = min(outer, o0 + chunk);
  const uint64_t per = (o1 > o0 ? o1 - o0 : 0) * q.inner;
  double d = 0.0, z = 0.0;
  for (uint64_t e = threadIdx.x; e < per; e += kBlock) {
    const uint64_t at = ((o0 + e / q.inner) * q.n_params + prm) * q.inner + e % q.inner;
    ste_bwd_f64(x[at], gy[at], p, true, d, z);
  }
  block_sum2(d, z);
  if (threadIdx.x == 0) {
    partial[(prm * gridDim.y + blockIdx.y) * 2] = d;
    partial[(prm * gridDim.y + blockIdx.y) * 2 + 1] = z;
  }
}

__global__ void fq_bwd_params_final_f64_k(const double* __restrict__ partial, uint32_t slices, tq_quantizer_f64 q,
                                          double* __restrict__ g_delta, double* __restrict__ g_zf) {
  const uint64_t prm = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (prm >= q.n_params) return;
  double d = 0.0, z = 0.0;
  for (uint32_t s = 0; s < slices; ++s) { d += partial[(prm * slices + s) * 2]; z += partial[(prm * slices + s) * 2 + 1]; }
  const double delta = q.delta[prm];
  const double pass = q.log_domain ? exp(delta) : (delta >= q.eps ? 1.0 : 0.0);
  g_delta[prm] = d * pass;
  if (g_zf != nullptr && !q.symmetric) {
    const QPD p = make_qpd(q, prm);
    const double zf = rint(q.zero_float[prm]);
    g_zf[prm] = (zf >= p.lo && zf <= p.hi) ? z : 0.0;
  }
}

// ------------------------------------------------------------------------------ min / max
// partial[(p * slices + s) * 2 + {0, 1}]: slice s of the `outer` index of parameter p.  inner == 1 (statistics
// along the last axis): a block covers 32 adjacent parameters x 8 outer rows per step -> 256-byte row segments.
__global__ __launch_bounds__(kBlock) void mm_f64_rows_k(const double* __restrict__ x, uint64_t outer, uint64_t n_params, uint64_t inner,
                                                        double* __restrict__ partial) {
  const uint64_t prm = blockIdx.x;
  const uint64_t chunk = (outer + gridDim.y - 1) / gridDim.y;
  const uint64_t o0 = (uint64_t)blockIdx.y * chunk, o1 = min(outer, o0 + chunk);
  const uint64_t per = (o1 > o0 ? o1 - o0 : 0) * inner;
  double mn = __builtin_inf(), mx = -__builtin_inf();
  for (uint64_t e = threadIdx.x; e < per; e += kBlock) {
    const double v = x[((o0 + e / inner) * n_params + prm) * inner + e % inner];
    mn = dmin_nanprop(mn, v);
    mx = dmax_nanprop(mx, v);
  }
  __shared__ double s_red[2][kBlock];
  s_red[0][threadIdx.x] = mn;
  s_red[1][threadIdx.x] = mx;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      s_red[0][threadIdx.x] = dmin_nanprop(s_red[0][threadIdx.x], s_red[0][threadIdx.x + o]);
      s_red[1][threadIdx.x] = dmax_nanprop(s_red[1][threadIdx.x], s_red[1][threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[(prm * gridDim.y + blockIdx.y) * 2] = s_red[0][0];
    partial[(prm * gridDim.y + blockIdx.y) * 2 + 1] = s_red[1][0];
  }
}

__global__ __launch_bounds__(kBlock) void mm_f64_cols_k(const double* __restrict__ x, uint64_t outer, uint64_t n_params,
                                                        double* __restrict__ partial) {
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;          // 32 columns x 8 rows
  const uint64_t prm = (uint64_t)blockIdx.x * 32 + cx;
  const uint64_t chunk = (outer + gridDim.y - 1) / gridDim.y;
  const uint64_t o0 = (uint64_t)blockIdx.y * chunk, o1 = min(outer, o0 + chunk);
  double mn = __builtin_inf(), mx = -__builtin_inf();
  if (prm < n_params)
    for (uint64_t o = o0 + ry; o < o1; o += 8) {
      const double v = x[o * n_params + prm];
      mn = dmin_nanprop(mn, v);
      mx = dmax_nanprop(mx, v);
    }
  __shared__ double s_red[2][8][32];
  s_red[0][ry][cx] = mn;
  s_red[1][ry][cx] = mx;
  __syncthreads();
  if (ry == 0 && prm < n_params) {
    for (int k = 1; k < 8; ++k) { mn = dmin_nanprop(mn, s_red[0][k][cx]); mx = dmax_nanprop(mx, s_red[1][k][cx]); }
    partial[(prm * gridDim.y + blockIdx.y) * 2] = mn;
    partial[(prm * gridDim.y + blockIdx.y) * 2 + 1] = mx;
  }
}

__global__ void mm_f64_final_k(const double* __restrict__ partial, uint32_t slices, uint64_t n_params, double* __restrict__ out_min,
                               double* __restrict__ out_max) {
  const uint64_t prm = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (prm >= n_params) return;
  double mn = __builtin_inf(), mx = -__builtin_inf();
  for (uint32_t s = 0; s < slices; ++s) {
    mn = dmin_nanprop(mn, partial[(prm * slices + s) * 2]);
    mx = dmax_nanprop(mx, partial[(prm * slices + s) * 2 + 1]);
  }
  out_min[prm] = mn;
  out_max[prm] = mx;
}

struct MmPlanF64 { bool cols; unsigned bx, slices; };
__host__ inline MmPlanF64 mm_plan_f64(uint64_t n, uint64_t n_params, uint64_t inner) {
  MmPlanF64 pl;
  const uint64_t outer = n / (n_params * inner);
  pl.cols = inner == 1 && n_params >= 32;
  pl.bx = (unsigned)(pl.cols ? ceil_div(n_params, 32) : n_params);
  // ~2048 blocks in flight, >= 4096 elements per block, never more slices than outer rows
  uint64_t s = std::max<uint64_t>(1, 2048 / std::max<uint64_t>(1, std::min<uint64_t>(pl.bx, 2048)));
  s = std::min<uint64_t>(s, std::max<uint64_t>(1, outer * inner * (pl.cols ? 32 : 1) / 4096));
  s = std::min<uint64_t>(std::min<uint64_t>(s, outer), 1024);
  pl.slices = (unsigned)std::max<uint64_t>(1, s);
  return pl;
}

// ------------------------------------------------------------------------------ estimator state / range -> parameters
__global__ void range_update_f64_k(int mode, const double* __restrict__ new_min, const double* __restrict__ new_max,
                                   double* __restrict__ cur_min, double* __restrict__ cur_max, uint64_t n, int initialised,
                                   double momentum, uint64_t n_groups, const int64_t* __restrict__ order) {
  extern __shared__ double s_gd[];   // [2][n_groups]
  const uint64_t gs = n_groups > 0 ? n / n_groups : 0;
  if (n_groups > 0) {
    for (uint64_t g = threadIdx.x; g < n_groups; g += blockDim.x) {
      double mn = __builtin_inf(), mx = -__builtin_inf();
      for (uint64_t k = 0; k < gs; ++k) {
        const uint64_t dim = order ? (uint64_t)order[g * gs + k] : g * gs + k;
        mn = dmin_nanprop(mn, new_min[dim]);
        mx = dmax_nanprop(mx, new_max[dim]);
      }
      s_gd[g] = mn;
      s_gd[n_groups + g] = mx;
    }
    __syncthreads();
  }
  const double om = 1.0 - momentum;
  for (uint64_t j = threadIdx.x; j < n; j += blockDim.x) {
    const uint64_t dim = (n_groups > 0 && order) ? (uint64_t)order[j] : j;
    const double a = n_groups > 0 ? s_gd[j / gs] : new_min[j];
    const double b = n_groups > 0 ? s_gd[n_groups + j / gs] : new_max[j];
    if (mode == TQ_EST_CURRENT || !initialised) { cur_min[dim] = a; cur_max[dim] = b; }
    else if (mode == TQ_EST_ALL) { cur_min[dim] = dmin_nanprop(cur_min[dim], a); cur_max[dim] = dmax_nanprop(cur_max[dim], b); }
    else { cur_min[dim] = om * a + momentum * cur_min[dim]; cur_max[dim] = om * b + momentum * cur_max[dim]; }
  }
}

__global__ void axis_ranges_f64_k(const double* __restrict__ new_min, const double* __restrict__ new_max, double* __restrict__ ranges,
                                  uint64_t n, int first) {
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
    const double r = new_max[j] - new_min[j];
    ranges[j] = first ? r : (0.1 * r + 0.9 * r);        // range_estimators.py:75-79
  }
}

__global__ void set_range_asym_f64_k(const double* __restrict__ x_min, const double* __restrict__ x_max, uint64_t n, int n_bits,
                                     double eps, int log_domain, double* __restrict__ delta, double* __restrict__ zero_float) {
  const double top = ldexp(1.0, n_bits) - 1.0;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
    const double lo = dmin_nanprop(x_min[j], 0.0);      // quantizers.py:258
    const double hi = dmax_nanprop(x_max[j], eps);      // :259
    const double d = (hi - lo) / top;                   // :276
    zero_float[j] = (-lo) / d;                          // :277
    delta[j] = log_domain ? log(d) : d;                 // :279-280
  }
}

__global__ void set_range_sym_f64_k(const double* __restrict__ x_min, const double* __restrict__ x_max, uint64_t n, int n_bits,
                                    double eps, int log_domain, double* __restrict__ delta, uint8_t* __restrict__ signed_flag) {
  __shared__ int s_neg;
  if (threadIdx.x == 0) s_neg = 0;
  __syncthreads();
  int neg = 0;
  for (uint64_t j = threadIdx.x; j < n; j += blockDim.x) neg |= (dmin_nanprop(x_min[j], 0.0) < 0.0) ? 1 : 0;
  if (neg) atomicOr(&s_neg, 1);
  __syncthreads();
  const bool sgn = s_neg != 0;
  if (threadIdx.x == 0) signed_flag[0] = sgn ? 1 : 0;
  const double top = ldexp(1.0, n_bits - (sgn ? 1 : 0)) - 1.0;     // :325-328
  for (uint64_t j = threadIdx.x; j < n; j += blockDim.x) {
    const double lo = dmin_nanprop(x_min[j], 0.0);
    const double hi = dmax_nanprop(x_max[j], eps);
    const double d = dmax_nanprop(fabs(lo), hi) / top;               // :338-339
    delta[j] = log_domain ? log(d) : d;
  }
}

// ------------------------------------------------------------------------------ MSE candidates
// loss[r, c] += sum over row r (or the whole tensor, reduce_rows) of (x - Q_c(x))^2; cand[c] = (scale, zp, lo, hi)
// fp32 values used in double arithmetic (torch's fp64 x / fp32-tensor scale promotion).  One block per (row, slice);
// a block keeps CT candidates' partial sums per thread, tiles over the candidate list, atomically adds its partials
// (fp64 atomics: the summation order across blocks is not fixed -- 1e-16-relative differences run to run).
constexpr int kCandTileF64 = 8;
__global__ __launch_bounds__(kBlock) void mse_cand_f64_k(const double* __restrict__ x, uint64_t rows, uint64_t row_len,
                                                         const float* __restrict__ cand, uint32_t n_cand, int reduce_rows,
                                                         double* __restrict__ loss) {
  const uint64_t row = blockIdx.x;
  const uint64_t chunk = (row_len + gridDim.y - 1) / gridDim.y;
  const uint64_t e0 = (uint64_t)blockIdx.y * chunk, e1 = min(row_len, e0 + chunk);
  const double* xr = x + row * row_len;
  for (uint32_t c0 = 0; c0 < n_cand; c0 += kCandTileF64) {
    QPD p[kCandTileF64];
    double acc[kCandTileF64];
#pragma unroll
    for (int t = 0; t < kCandTileF64; ++t) {
      const uint32_t c = min(c0 + t, n_cand - 1);
      p[t] = QPD{(double)cand[4 * c], (double)cand[4 * c + 1], (double)cand[4 * c + 2], (double)cand[4 * c + 3]};
      acc[t] = 0.0;
    }
    for (uint64_t e = e0 + threadIdx.x; e < e1; e += kBlock) {
      const double v = xr[e];
#pragma unroll
      for (int t = 0; t < kCandTileF64; ++t) {
        const double dq = v - qd_dequant(qd_index(v, p[t]), p[t]);
        acc[t] += dq * dq;
      }
    }
#pragma unroll
    for (int t = 0; t < kCandTileF64; t += 2) {
      double a = acc[t], b = acc[t + 1];
      block_sum2(a, b);
      if (threadIdx.x == 0) {
        double* dst = loss + (reduce_rows ? 0 : row) * n_cand;
        if (c0 + t < n_cand) atomicAdd(dst + c0 + t, a);
        if (c0 + t + 1 < n_cand) atomicAdd(dst + c0 + t + 1, b);
      }
    }
  }
}

static int check_qd(const tq_quantizer_f64* q, uint64_t n, const char* who) {
  TQ_REQUIRE(q != nullptr && q->delta != nullptr, "%s: quantizer / delta is NULL", who);
  TQ_REQUIRE(q->n_bits >= 1 && q->n_bits <= 52, "%s: n_bits=%d outside 1..52", who, q->n_bits);
  TQ_REQUIRE(q->symmetric || q->zero_float != nullptr, "%s: asymmetric quantizer without zero_float", who);
  TQ_REQUIRE(q->n_params >= 1 && q->inner >= 1, "%s: n_params / inner must be >= 1", who);
  TQ_REQUIRE(q->n_params == 1 || n % (q->n_params * q->inner) == 0, "%s: n=%llu is not a multiple of n_params*inner=%llu", who,
             (unsigned long long)n, (unsigned long long)(q->n_params * q->inner));
  return TQ_OK;
}

}  // namespace tq

using namespace tq;

extern "C" int tq_fake_quant_fwd_f64(const double* x, double* y, double* idx, uint64_t n, const tq_quantizer_f64* q,
                                     tq_stream_t stream) {
  if (n == 0) return TQ_OK;
  TQ_REQUIRE(x != nullptr && (y != nullptr || idx != nullptr), "tq_fake_quant_fwd_f64: NULL pointer");
  if (int e = check_qd(q, n, "tq_fake_quant_fwd_f64")) return e;
  const bool vec = n % 2 == 0 && aligned16(x) && (y == nullptr || aligned16(y)) && (idx == nullptr || aligned16(idx));
  const unsigned grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>(ceil_div(vec ? n / 2 : n, kBlock), 1), kMaxGrid * 4);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (vec) hipLaunchKernelGGL((fq_f64_k<true>), dim3(grid), dim3(kBlock), 0, st, x, y, idx, n, *q);
  else     hipLaunchKernelGGL((fq_f64_k<false>), dim3(grid), dim3(kBlock), 0, st, x, y, idx, n, *q);
  return check_launch("fq_f64_k");
}

extern "C" size_t tq_fake_quant_bwd_f64_workspace_bytes(uint64_t n, uint64_t n_params, uint64_t inner) {
  if (n_params == 0 || inner == 0) return 0;
  const uint64_t outer = n / (n_params * inner);
  return (size_t)n_params * bwd_slices_f64(outer, inner) * 2 * sizeof(double);
}

extern "C" int tq_fake_quant_bwd_f64(const double* x, const double* grad_y, double* grad_x, double* g_delta, double* g_zero_float,
                                     uint64_t n, const tq_quantizer_f64* q, void* workspace, size_t workspace_bytes,
                                     tq_stream_t stream) {
  if (n == 0) return TQ_OK;
  TQ_REQUIRE(x && grad_y && grad_x, "tq_fake_quant_bwd_f64: NULL pointer");
  if (int e = check_qd(q, n, "tq_fake_quant_bwd_f64")) return e;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>(ceil_div(n, kBlock), 1), kMaxGrid * 4);
  hipLaunchKernelGGL(fq_bwd_f64_k, dim3(grid), dim3(kBlock), 0, st, x, grad_y, grad_x, n, *q);
  if (int e = check_launch("fq_bwd_f64_k")) return e;
  if (g_delta == nullptr) return TQ_OK;
  TQ_REQUIRE(q->n_params <= 65535u * 32u, "tq_fake_quant_bwd_f64: too many parameters");
  const uint64_t outer = n / (q->n_params * q->inner);
  const unsigned slices = bwd_slices_f64(outer, q->inner);
  TQ_REQUIRE(workspace && workspace_bytes >= (size_t)q->n_params * slices * 2 * sizeof(double),
             "tq_fake_quant_bwd_f64: workspace too small");
  TQ_REQUIRE(q->n_params <= 2147483647u, "tq_fake_quant_bwd_f64: n_params too large");
  double* ws = static_cast<double*>(workspace);
  hipLaunchKernelGGL(fq_bwd_params_f64_k, dim3((unsigned)q->n_params, slices), dim3(kBlock), 0, st, x, grad_y, outer, *q, ws);
  if (int e = check_launch("fq_bwd_params_f64_k")) return e;
  hipLaunchKernelGGL(fq_bwd_params_final_f64_k, dim3((unsigned)ceil_div(q->n_params, 256)), dim3(256), 0, st, ws, slices, *q, g_delta,
                     g_zero_float);
  return check_launch("fq_bwd_params_final_f64_k");
}

extern "C" size_t tq_minmax_f64_workspace_bytes(uint64_t n, uint64_t n_params, uint64_t inner) {
  if (n_params == 0 || inner == 0) return 0;
  return (size_t)n_params * mm_plan_f64(n, n_params, inner).slices * 2 * sizeof(double);
}

extern "C" int tq_minmax_f64(const double* x, uint64_t n, uint64_t n_params, uint64_t inner, double* out_min, double* out_max,
                             void* workspace, size_t workspace_bytes, tq_stream_t stream) {
  TQ_REQUIRE(x && out_min && out_max && n > 0, "tq_minmax_f64: bad argument");
  TQ_REQUIRE(n_params >= 1 && inner >= 1 && n % (n_params * inner) == 0, "tq_minmax_f64: n is not a multiple of n_params * inner");
  const uint64_t outer = n / (n_params * inner);
  const MmPlanF64 pl = mm_plan_f64(n, n_params, inner);
  const unsigned slices = pl.slices;
  TQ_REQUIRE(n_params <= 2147483647u, "tq_minmax_f64: too many parameters");
  TQ_REQUIRE(workspace && workspace_bytes >= (size_t)n_params * slices * 2 * sizeof(double), "tq_minmax_f64: workspace too small");
  double* ws = static_cast<double*>(workspace);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (pl.cols) hipLaunchKernelGGL(mm_f64_cols_k, dim3(pl.bx, slices), dim3(kBlock), 0, st, x, outer, n_params, ws);
  else         hipLaunchKernelGGL(mm_f64_rows_k, dim3(pl.bx, slices), dim3(kBlock), 0, st, x, outer, n_params, inner, ws);
  if (int e = check_launch("tq_minmax_f64 partial")) return e;
  hipLaunchKernelGGL(mm_f64_final_k, dim3((unsigned)ceil_div(n_params, 256)), dim3(256), 0, st, ws, slices, n_params, out_min, out_max);
  return check_launch("tq_minmax_f64 final");
}

extern "C" int tq_range_update_f64(int mode, const double* new_min, const double* new_max, double* cur_min, double* cur_max,
                                   uint64_t n, int initialised, double momentum, uint64_t n_groups, const int64_t* order,
                                   tq_stream_t stream) {
  TQ_REQUIRE(new_min && new_max && cur_min && cur_max, "tq_range_update_f64: NULL pointer");
  TQ_REQUIRE(mode >= TQ_EST_CURRENT && mode <= TQ_EST_RUNNING, "tq_range_update_f64: bad mode %d", mode);
  TQ_REQUIRE(n > 0, "tq_range_update_f64: n == 0");
  TQ_REQUIRE(n_groups == 0 || n % n_groups == 0, "tq_range_update_f64: n %% n_groups != 0");
  TQ_REQUIRE(n_groups * 2 * sizeof(double) <= 64 * 1024, "tq_range_update_f64: too many groups");
  hipLaunchKernelGGL(range_update_f64_k, dim3(1), dim3(n >= 256 ? 1024 : 256), n_groups * 2 * sizeof(double),
                     static_cast<hipStream_t>(stream), mode, new_min, new_max, cur_min, cur_max, n, initialised, momentum, n_groups,
                     order);
  return check_launch("tq_range_update_f64");
}

extern "C" int tq_axis_ranges_f64(const double* new_min, const double* new_max, double* ranges, uint64_t n, int first,
                                  tq_stream_t stream) {
  TQ_REQUIRE(new_min && new_max && ranges && n > 0, "tq_axis_ranges_f64: bad argument");
  hipLaunchKernelGGL(axis_ranges_f64_k, dim3((unsigned)std::min<uint64_t>(ceil_div(n, 256), 1024)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), new_min, new_max, ranges, n, first);
  return check_launch("tq_axis_ranges_f64");
}

extern "C" int tq_set_range_asym_f64(const double* x_min, const double* x_max, uint64_t n, int n_bits, double eps, int log_domain,
                                     double* delta, double* zero_float, tq_stream_t stream) {
  TQ_REQUIRE(x_min && x_max && delta && zero_float && n > 0, "tq_set_range_asym_f64: bad argument");
  TQ_REQUIRE(n_bits >= 1 && n_bits <= 52, "tq_set_range_asym_f64: n_bits=%d", n_bits);
  hipLaunchKernelGGL(set_range_asym_f64_k, dim3((unsigned)std::min<uint64_t>(ceil_div(n, 256), 1024)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x_min, x_max, n, n_bits, eps, log_domain, delta, zero_float);
  return check_launch("tq_set_range_asym_f64");
}

extern "C" int tq_set_range_sym_f64(const double* x_min, const double* x_max, uint64_t n, int n_bits, double eps, int log_domain,
                                    double* delta, uint8_t* signed_flag, tq_stream_t stream) {
  TQ_REQUIRE(x_min && x_max && delta && signed_flag && n > 0, "tq_set_range_sym_f64: bad argument");
  TQ_REQUIRE(n_bits >= 1 && n_bits <= 52, "tq_set_range_sym_f64: n_bits=%d", n_bits);
  hipLaunchKernelGGL(set_range_sym_f64_k, dim3(1), dim3(n >= 256 ? 1024 : 256), 0, static_cast<hipStream_t>(stream), x_min, x_max, n,
                     n_bits, eps, log_domain, delta, signed_flag);
  return check_launch("tq_set_range_sym_f64");
}

extern "C" int tq_mse_candidates_f64(const double* x, uint64_t rows, uint64_t row_len, const float* cand, uint64_t n_cand,
                                     int reduce_rows, double* loss, tq_stream_t stream) {
  if (rows == 0 || row_len == 0 || n_cand == 0) return TQ_OK;
  TQ_REQUIRE(x && cand && loss, "tq_mse_candidates_f64: NULL pointer");
  TQ_REQUIRE(rows <= 2147483647u && n_cand <= 0xffffffffu, "tq_mse_candidates_f64: shape too large");
  const unsigned slices = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(row_len / 2048, std::max<uint64_t>(1, 2048 / rows)));
  hipLaunchKernelGGL(mse_cand_f64_k, dim3((unsigned)rows, std::min(slices, 65535u)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                     x, rows, row_len, cand, (uint32_t)n_cand, reduce_rows, loss);
  return check_launch("mse_cand_f64_k");
}
