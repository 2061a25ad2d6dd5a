/* TEST INFRASTRUCTURE (oracle/): LayerNorm with its statistics summed in the order of the fused tail kernel
 * (transformer-quantization_amd/csrc/tq_fused_ln.hip, res_ln_body), so that the f2 chain
 *     Q_ln( LayerNorm( Q_res( Q_dense(a) + r ) ) )          reference models/quantized_bert.py:238-248, 264-280
 * can be compared with the kernel at ZERO tolerance.  torch.nn.functional.layer_norm (the reference's LayerNorm,
 * quantization/autoquant_utils.py run_forward) leaves the summation order of its fp32 statistics to the backend -- CPU,
 * GPU and this kernel all differ -- which is why the kernel's contract against the reference is a tolerance; THIS file
 * pins what the kernel computes, operation by operation (each one a single correctly rounded fp32 operation):
 *
 *   row of d = LPR * NV * V values, V = 4 (fp32 storage) or 8 (bf16 / fp16): lane l owns the 16-byte vectors
 *   v * LPR + l (v = 0 .. NV-1), each H = V / 2 pairs (x, y) = elements (2 j, 2 j + 1).
 *   sum:   two packed accumulators per lane, pair j of every vector goes to s2 (j even) or s2b (j odd) in (v, j) order;
 *          s2 += s2b; lane value = s2.x + s2.y; lanes combined by a butterfly over lane ^ 1, ^ 2, the mirror steps
 *          (other quad of 8, other half of 16), ^ 16, ^ 32 == a balanced binary tree over the lanes in natural order;
 *          mean = total * RN(1 / d).
 *   var:   c = u - mean; two packed fma accumulators, pair (v * H + j) even -> ssa = fma(c, c, ssa), odd -> ssb;
 *          ss = ssa + ssb; lane value = ss.x + ss.y; same tree; rstd = 1 / sqrt(total * RN(1 / d) + eps).
 *   out:   ((u - mean) * rstd) * w + b.
 * The mapping d -> (LPR, NV) is launch_res_ln's table, passed in by the caller (oracle/ln_sum.py).
 */
#include <math.h>
#include <stdint.h>

static float tree_sum(const float* lane, int lpr) {
  float t[64] = {0};
  for (int i = 0; i < lpr; ++i) t[i] = lane[i];
  for (int w = 1; w < lpr; w <<= 1)
    for (int i = 0; i < lpr; i += 2 * w) t[i] = t[i] + t[i + w];
  return t[0];
}

void tq_oracle_layernorm_kernel_order(const float* u, float* out, int64_t rows, int lpr, int nv, int V, const float* w,
                                      const float* b, float eps) {
  const int H = V / 2;
  const int64_t d = (int64_t)lpr * nv * V;
  const float inv_d = 1.0f / (float)d;
  for (int64_t r = 0; r < rows; ++r) {
    const float* x = u + r * d;
    float lane[64];
    for (int l = 0; l < lpr; ++l) {
      float s2x = 0.f, s2y = 0.f, sbx = 0.f, sby = 0.f;
      for (int v = 0; v < nv; ++v)
        for (int j = 0; j < H; ++j) {
          const float* p = x + ((int64_t)v * lpr + l) * V + 2 * j;
          if (j & 1) { sbx = sbx + p[0]; sby = sby + p[1]; }
          else { s2x = s2x + p[0]; s2y = s2y + p[1]; }
        }
      s2x = s2x + sbx;
      s2y = s2y + sby;
      lane[l] = s2x + s2y;
    }
    const float mean = tree_sum(lane, lpr) * inv_d;
    for (int l = 0; l < lpr; ++l) {
      float ax = 0.f, ay = 0.f, bx = 0.f, by = 0.f;
      for (int v = 0; v < nv; ++v)
        for (int j = 0; j < H; ++j) {
          const float* p = x + ((int64_t)v * lpr + l) * V + 2 * j;
          const float cx = p[0] - mean, cy = p[1] - mean;
          if ((v * H + j) & 1) { bx = fmaf(cx, cx, bx); by = fmaf(cy, cy, by); }
          else { ax = fmaf(cx, cx, ax); ay = fmaf(cy, cy, ay); }
        }
      ax = ax + bx;
      ay = ay + by;
      lane[l] = ax + ay;
    }
    const float var = tree_sum(lane, lpr) * inv_d + eps;
    const float rstd = 1.0f / sqrtf(var);
    for (int64_t e = 0; e < d; ++e) out[r * d + e] = ((x[e] - mean) * rstd) * w[e] + b[e];
  }
}
