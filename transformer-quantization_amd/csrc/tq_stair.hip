// Activation + output quantizer of an integer Linear as a staircase table (round 4; docs/history/DESIGN_rounds_1-4.md section 3.4).
//
// The fused Linear + GELU + quantizer epilogue (reference hijacker.py:66-116 behind autoquant_utils.py:16-21: F.linear ->
// nn.GELU() -> activation quantizer) spent ~18 VALU instructions (most of them packed, i.e. double-cost) per output on the
// erf fit and the quantizer's exact quotient, and on CDNA4 that VALU time ADDS to the matrix-core time
// (tools/tuning/mfma_valu_overlap.hip).  But
//     h(v) = clamp(rne(RN32(GELU(v)) / scale) + zp, lo, hi) - zp
// is a step function of the fp32 pre-activation v with at most 255 steps, so it can be tabulated exactly.  This file
// builds the table on the device (no host read of the range buffers: hipGraph-capturable like every other launch):
//   * RN32(GELU(v)) is the CORRECTLY ROUNDED fp32 value of x/2 (1 + erf(x / sqrt 2)) (float64 evaluation, then narrowed) --
//     what nn.GELU() approximates to ~1 ulp -- followed by the reference's own fp32 quantizer arithmetic (q_index: IEEE
//     division, rne, clamp; quantizers.py:184-185).  oracle/tq_int_oracle.c evaluates exactly this per element, with
//     no table, and tests/test_int_oracle.py compares the kernel with it at zero tolerance.
//   * bins: uniform in v between v0 (where |GELU| has fallen below a quarter grid step on the negative side) and v1 (just
//     beyond the quantizer's upper clamp); the fp32 interval of every bin is derived from `stair_bin` itself by bisection
//     over fp32 ordinals; one thread per bin checks that h is constant or makes exactly ONE unit step inside its interval
//     (GELU is unimodal: monotone on either side of its minimum at v = -0.7518; the bin that holds the minimum must be
//     flat) and locates the step's first fp32 value by bisection.
//   * if any bin fails (grid too fine for the bin count: scale below ~0.01 at 768 bins) the header says so and consumers
//     keep the arithmetic epilogue; the decision is made on the device and read by the consuming kernel.
#include "tq_device.h"
#include "tq_host.h"

namespace tq {

constexpr int kStairMaxBins = 2048;
constexpr double kGeluArgmin = -0.75179162867227847;     // GELU'(v) = 0

__device__ __forceinline__ uint32_t ord_of(float f) {     // monotone: ascending keys == ascending finite values
  const uint32_t b = f32_to_bits(f);
  return (b & 0x80000000u) ? ~b : (b ^ 0x80000000u);
}
__device__ __forceinline__ float of_ord(uint32_t k) { return bits_to_f32((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }

__device__ __forceinline__ double act64(double v, int act) {
  switch (act) {
    case TQ_ACT_GELU: return 0.5 * v * (1.0 + erf(v * 0.70710678118654752440));
    case TQ_ACT_RELU: return v > 0.0 ? v : 0.0;
    default: return v;
  }
}
// h(v): the reference quantizer applied to the correctly rounded activation
__device__ __forceinline__ float stair_h(float v, int act, const QP& qp) { return q_index((float)act64((double)v, act), qp) - qp.zp; }

__global__ __launch_bounds__(1024) void stair_build_k(tq_quantizer q, int act, uint32_t nb, float* __restrict__ table) {
  const QP qp = make_qp(q, 0);
  // geometry (every thread derives the same scalars)
  const double s = (double)qp.scale;
  double v1 = ((double)qp.hi - (double)qp.zp + 0.5) * s;          // act(v) >= this saturates the grid
  v1 = (v1 > 0.0 ? v1 : 0.0) + 0.25;                               // GELU(v) > v - 0.17
  double v0 = -1.0;
  if (act == TQ_ACT_GELU) {
    // largest v <= argmin with |GELU(v)| <= s / 4 (h is constant below it); at least one unit left of the minimum
    double a = -40.0, b = kGeluArgmin;
    for (int it = 0; it < 60; ++it) {
      const double m = 0.5 * (a + b);
      if (-act64(m, act) <= 0.25 * s) a = m; else b = m;
    }
    v0 = a < -1.75 ? a : -1.75;
  } else {
    v0 = ((double)qp.lo - (double)qp.zp - 0.5) * s;
    v0 = (v0 < 0.0 ? v0 : 0.0) - 0.25;
  }
  const float inv_w = (float)((double)nb / (v1 - v0));
  const float c0 = (float)(-v0 * (double)inv_w);
  const float nbm1 = (float)(nb - 1);
  const bool geom_ok = isfinite(inv_w) && isfinite(c0) && inv_w > 0.0f && qp.hi - qp.lo <= 255.0f && fabsf(qp.lo - qp.zp) <= 256.0f &&
                       fabsf(qp.hi - qp.zp) <= 256.0f && qp.scale > 0.0f && qp.zp == qp.zp;
  const uint32_t k_first = ord_of(-3.4028234663852886e38f), k_last = ord_of(3.4028234663852886e38f);
  const float vmin32 = (float)kGeluArgmin;
  int bad = geom_ok ? 0 : 1;
  uint32_t* out = reinterpret_cast<uint32_t*>(table + 4);
  for (uint32_t k = threadIdx.x; k < nb && geom_ok; k += blockDim.x) {
    // fp32 interval of bin k: [smallest ordinal with bin >= k, smallest ordinal with bin >= k + 1)
    auto first_with_bin_ge = [&](uint32_t want) -> uint32_t {      // want in [1, nb - 1]; bins are monotone in v
      uint32_t lo = k_first, hi = k_last;                          // bin(lo) < want <= bin(hi) (bin(k_last) = nb - 1)
      if (stair_bin(of_ord(lo), inv_w, c0, nbm1) >= want) return lo;
      while (hi - lo > 1) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (stair_bin(of_ord(mid), inv_w, c0, nbm1) >= want) hi = mid; else lo = mid;
      }
      return hi;
    };
    const uint32_t klo = k == 0 ? k_first : first_with_bin_ge(k);
    const uint32_t khi = k == nb - 1 ? k_last : first_with_bin_ge(k + 1) - 1;
    float T = 3.4028234663852886e38f, hl = 0.0f, hr = 0.0f;
    if (klo <= khi) {
      const float a = of_ord(klo), b = of_ord(khi);
      hl = stair_h(a, act, qp);
      hr = stair_h(b, act, qp);
      if (act == TQ_ACT_GELU && a <= vmin32 && vmin32 <= b) {
        // the bin around the minimum is V-shaped: exact only if it is flat
        if (!(hl == hr && stair_h(vmin32, act, qp) == hl)) bad = 1;
        T = a;
      } else if (hl == hr) {
        T = a;                                                     // no step: either half of `packed` is right
      } else if (fabsf(hr - hl) == 1.0f) {
        uint32_t lo = klo, hi = khi;                               // h(lo) = hl, h(hi) = hr, monotone in between
        while (hi - lo > 1) {
          const uint32_t mid = lo + ((hi - lo) >> 1);
          if (stair_h(of_ord(mid), act, qp) == hr) hi = mid; else lo = mid;
        }
        T = of_ord(hi);
      } else {
        bad = 1;                                                   // two or more steps in one bin
      }
    }
    out[2 * k] = f32_to_bits(T);
    out[2 * k + 1] = (f32_to_bits(hr) & 0xffff0000u) | (f32_to_bits(hl) >> 16);
  }
  const int any_bad = __syncthreads_or(bad);
  if (threadIdx.x == 0) {
    table[0] = inv_w;
    table[1] = c0;
    table[2] = nbm1;
    table[3] = any_bad ? 0.0f : 1.0f;
  }
}

}  // namespace tq

using namespace tq;

extern "C" size_t tq_act_stair_bytes(uint32_t n_bins) { return sizeof(StairHdr) + (size_t)n_bins * 8; }

extern "C" int tq_act_stair_build(int activation, const tq_quantizer* q_out, uint32_t n_bins, void* table, size_t table_bytes,
                                  tq_stream_t stream) {
  TQ_REQUIRE(q_out && table, "tq_act_stair_build: NULL pointer");
  TQ_REQUIRE(activation == TQ_ACT_NONE || activation == TQ_ACT_RELU || activation == TQ_ACT_GELU,
             "tq_act_stair_build: activation %d has no staircase (none, relu, gelu)", activation);
  TQ_REQUIRE(n_bins >= 64 && n_bins <= (uint32_t)kStairMaxBins, "tq_act_stair_build: 64..%d bins, got %u", kStairMaxBins, n_bins);
  TQ_REQUIRE(table_bytes >= tq_act_stair_bytes(n_bins), "tq_act_stair_build: table too small");
  TQ_REQUIRE(aligned16(table), "tq_act_stair_build: 16-byte alignment required");
  if (int e = check_quantizer(q_out, 1, "tq_act_stair_build")) return e;
  TQ_REQUIRE(q_out->n_params == 1 && q_out->n_bits >= 1 && q_out->n_bits <= 8, "tq_act_stair_build: per-tensor <= 8-bit quantizer only");
  hipLaunchKernelGGL(stair_build_k, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), *q_out, activation, n_bins,
                     static_cast<float*>(table));
  return check_launch("stair_build_k");
}
