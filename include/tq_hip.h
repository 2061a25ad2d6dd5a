/*
 * tq_hip.h -- C ABI of libtq_hip.so: the MI355X (gfx950) fake-quantization hot path.
 *
 * The reference (Qualcomm-AI-research/transformer-quantization) is pure Python/PyTorch and has
 * no FFI of its own; the interface each entry point replaces is therefore a Python method of
 * the reference's quantization/ package, cited per function as file:line (paths relative to
 * the upstream repository root).  INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add to call these from the reference's own classes.
 *
 * Conventions
 *   - Every pointer named x/y/idx/delta/... is a DEVICE pointer (HBM) unless the comment says
 *     "host".  The library never allocates, frees or synchronises: the caller supplies every
 *     buffer, including workspaces whose size is returned by the *_workspace_bytes() helpers.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  All work
 *     is enqueued on it and is re-entrant per stream.
 *   - Return value: 0 on success, a negative TQ_E* code otherwise; tq_last_error() returns a
 *     thread-local, human readable description of the last failure.  No C++ exception crosses
 *     the ABI.
 *   - Tensor layout: contiguous row-major.  Quantization parameters are selected per element
 *     i (flat index) as  p = (i / inner) % n_params :
 *         per-tensor                n_params = 1            (inner ignored)
 *         per-embedding / axis=-1   n_params = d, inner = 1
 *         per-channel (dim 0)       n_params = C, inner = numel / C
 *         any other axis            n_params = shape[axis], inner = prod(shape[axis+1:])
 *   - Numerics contract: all arithmetic in IEEE fp32 with true division
 *     and round-half-to-even; bf16/fp16 storage is widened to fp32 in registers and the
 *     dequantised value rounded (RNE) on store.  Integer indices are bit-exact with the
 *     reference's CPU path; clamp propagates NaN like torch.clamp.
 */
#ifndef TQ_HIP_H
#define TQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 6): tq_embeddings_layernorm_quant_fwd takes `bad_ids` (a signature change of an existing entry point; additions
 * alone had not moved the number). */
#define TQ_ABI_VERSION 2

/* storage dtype of x / y */
enum { TQ_F32 = 0, TQ_BF16 = 1, TQ_F16 = 2 };
/* storage dtype of the optional integer-index output */
enum { TQ_IDX_NONE = 0, TQ_IDX_F32 = 1, TQ_IDX_I8 = 2, TQ_IDX_U8 = 3, TQ_IDX_I16 = 4,
       TQ_IDX_I32 = 5, TQ_IDX_I8_M128 = 6 /* int8(index - 128): operand of tq_linear_i8_fwd */ };
/* error codes */
enum { TQ_OK = 0, TQ_EINVAL = -1, TQ_ELAUNCH = -2, TQ_EWORKSPACE = -3, TQ_EUNSUPPORTED = -4 };
/* range-estimator update rules for tq_range_update */
enum { TQ_EST_CURRENT = 0, TQ_EST_ALL = 1, TQ_EST_RUNNING = 2 };
/* AdaRound relaxations (quantization/adaround/utils.py:63-76) */
enum { TQ_ADA_SIGMOID = 0, TQ_ADA_HARD_SIGMOID = 1, TQ_ADA_SIGMOID_TEMP = 2 };

typedef void* tq_stream_t;

/* Quantizer description shared by the entry points below.  Mirrors the state of
 * AsymmetricUniformQuantizer / SymmetricUniformQuantizer (quantization/quantizers.py:96-107,
 * 306-308): the raw `_delta`, `_zero_float` and `_signed` buffers stay on the device and are
 * turned into (scale, zero_point, int_min, int_max) inside the kernels, exactly as the
 * `scale` / `zero_point` / `int_min` / `int_max` properties do (:132-153, :321-332), so no host
 * synchronisation is needed to launch.                                                        */
typedef struct tq_quantizer {
  const float*   delta;        /* [n_params] raw _delta (log-domain value if log_domain)      */
  const float*   zero_float;   /* [n_params] raw _zero_float, NULL for symmetric               */
  const uint8_t* signed_flag;  /* 1 byte (_signed), symmetric only; NULL = unsigned            */
  int32_t        n_bits;       /* 1..24                                                        */
  int32_t        symmetric;    /* 0 = asymmetric_uniform, 1 = symmetric_uniform                */
  int32_t        log_domain;   /* scale_domain == 'log' (scale = exp(delta))                   */
  float          eps;          /* quantizers.py:96 (default 1e-8)                              */
  uint64_t       n_params;     /* see layout note above                                        */
  uint64_t       inner;
} tq_quantizer;

int tq_abi_version(void);
const char* tq_last_error(void);

/* ---- K1/K2/K3: fused quantize -> clip -> dequantize ---------------------------------------
 * Replaces AsymmetricUniformQuantizer.forward / to_integer_forward and the symmetric subclass
 * (quantization/quantizers.py:172-211, 291-349): 6 ATen kernels become one.
 * y (same dtype as x) and idx are both optional but not both NULL.                           */
int tq_fake_quant_fwd(const void* x, void* y, void* idx, int idx_dtype, uint64_t n, int dtype,
                      const tq_quantizer* q, tq_stream_t stream);

/* The same for MANY independent tensors in one launch (40 per launch; more are split): what a model does with its weight
 * tensors once per range state -- the reference quantizes and caches them one by one in eval mode (quantization/
 * hijacker.py:52-64, base_quantized_classes.py:62-75).  Every item has its own quantizer (per-tensor, or n_params rows
 * of `inner` elements with inner a multiple of a 16-byte vector: per-output-channel weights); all items share `dtype`.
 * `items` is a HOST array, read during the call (the table travels as a kernel argument: no upload, hipGraph-capturable).
 * Bit-identical to n_items calls of tq_fake_quant_fwd.                                                                */
typedef struct tq_fq_item {
  const void*  x;
  void*        y;            /* same dtype and size as x */
  uint64_t     n;            /* elements; 0 = skipped    */
  tq_quantizer q;
} tq_fq_item;
int tq_fake_quant_multi_fwd(const tq_fq_item* items, uint32_t n_items, int dtype, tq_stream_t stream);

/* Fused NoNorm + output quantizer (MobileBERT; reference models/quantized_mobilebert.py:58-72,
 * QuantNoNorm.forward followed by quantize_activations): y = Q(x * w[col] + b[col]) for x viewed
 * as [n / d, d]; w, b fp32 [d] (the already fake-quantized affine parameters); per-tensor output
 * quantizer.  mul and add are separate fp32 operations like the reference's `x * weight + bias`. */
int tq_affine_fake_quant_fwd(const void* x, const float* w, const float* b, void* y,
                             int8_t* y_idx /* optional int8(index - 128) of y for a following integer Linear, or NULL */,
                             uint64_t n, uint64_t d, int dtype, const tq_quantizer* q, tq_stream_t stream);

/* (f2) Fused tail of BertSelfOutput / BertOutput with fixed ranges (reference
 * models/quantized_bert.py:238-248, 264-280):
 *     y = Q_out( layer_norm( Q_sum( Q_dense(dense_out) + residual ); ln_weight, ln_bias, eps ) )
 * for [rows, d] tensors: 2 reads + 1 write instead of 5 + 4.  Each quantizer is per-tensor and may be
 * NULL (= identity).  ln_weight / ln_bias: fp32 [d], already fake-quantized.  Statistics (mean,
 * biased centred variance) are fp32 two-pass over the row held in registers.  Supported row lengths:
 * d / (16-byte vector width) in {16,32,64,96,128,192,256,384,512,768}.                              */
int tq_residual_layernorm_quant_fwd(const void* dense_out, const void* residual, void* y,
                                    int8_t* y_idx /* optional int8(index - 128) of y, or NULL */,
                                    uint64_t rows, uint64_t d, int dtype,
                                    const tq_quantizer* q_dense, const tq_quantizer* q_sum,
                                    const float* ln_weight, const float* ln_bias, float ln_eps,
                                    const tq_quantizer* q_out, tq_stream_t stream);

/* The same tail with MobileBERT's NoNorm (element-wise affine, no statistics) in place of LayerNorm
 * (reference models/quantized_mobilebert.py:58-72 with :287-304, :330-352):
 *     y = Q_out( Q_sum( Q_dense(dense_out) + residual ) * weight + bias )
 * weight / bias: fp32 [d], already fake-quantized; mul and add are separate fp32 operations.       */
int tq_residual_nonorm_quant_fwd(const void* dense_out, const void* residual, void* y, int8_t* y_idx,
                                 uint64_t rows, uint64_t d, int dtype, const tq_quantizer* q_dense,
                                 const tq_quantizer* q_sum, const float* weight, const float* bias,
                                 const tq_quantizer* q_out, tq_stream_t stream);

/* BERT's embedding block with fixed ranges as ONE launch (reference models/quantized_bert.py:75-111, QuantizedBertEmbeddings:
 * three QuantEmbedding look-ups, two activation quantizers, LayerNorm + its output quantizer -- 13 launches of the layered
 * route):
 *     y = Q_out( LayerNorm( Q_sum2( Q_sum1( word[word_ids] + type[type_ids] ) + pos[pos_ids] ) ) )
 * The tables are fp32 [*_rows, d], already fake-quantized (the eval-mode parameter cache of QuantEmbedding); the ids are
 * int64 [rows] on the device.  An id outside its table -- where torch's CPU F.embedding raises IndexError -- never reads
 * out of bounds: the output row is NaN as a whole (its y_idx bytes unspecified) and, when bad_ids is given, *bad_ids is set
 * to 1 (a 4-byte flag the kernel can write and the HOST can read without synchronising, e.g. pinned host memory; never
 * cleared by the library).  Same element arithmetic and summation order as tq_residual_layernorm_quant_fwd with
 * dense_out = word + type (one fp32 addition) and residual = pos; d as there (fp32 rows).                              */
int tq_embeddings_layernorm_quant_fwd(const float* word_table, uint64_t word_rows, const int64_t* word_ids,
                                      const float* type_table, uint64_t type_rows, const int64_t* type_ids,
                                      const float* pos_table, uint64_t pos_rows, const int64_t* pos_ids, float* y,
                                      int8_t* y_idx /* optional */, uint64_t rows, uint64_t d,
                                      const tq_quantizer* q_sum1, const tq_quantizer* q_sum2, const float* ln_weight,
                                      const float* ln_bias, float ln_eps, const tq_quantizer* q_out,
                                      int32_t* bad_ids /* optional */, tq_stream_t stream);

/* (f3) Fused integer Linear + bias + activation + output quantizer on the i8 matrix cores.
 * Replaces QuantizationHijacker.forward for a Linear with fixed ranges (quantization/hijacker.py:
 * 66-116 + autoquant_utils.py:16-21):  y = Q_out( act( F.linear(Q_x(x), Q_w(W), b) ) ), evaluated
 * exactly on the integer grids:
 *     y_pre[m,n] = s_x s_w[n] ( sum_k x_idx[m,k] w_idx[n,k] + (128 - z_x) w_rowsum[n] ) + b[n]
 *   x_idx   int8 [M, K]  activation indices minus 128 (tq_fake_quant_fwd with TQ_IDX_I8_M128);
 *                         the input quantizer is asymmetric per-tensor, <= 8 bits (x_delta,
 *                         x_zero_float: its raw device buffers)
 *   w_idx   int8 [N, K]  weight indices of a symmetric quantizer (TQ_IDX_I8), w_rowsum int32 [N]
 *                         their row sums (tq_rowsum_i8, once per weight), w_delta [1] or [N]
 *   activation 0 none, 1 ReLU, 2 GELU (erf), 3 Tanh;  q_out NULL or a per-tensor quantizer
 *   y       [M, N] fp32 or bf16, or NULL with y_idx given (index-only output: the consumer is another integer
 *           Linear, e.g. BERT's intermediate -> output pair; saves the fp32 store, 4/5 of the kernel's HBM writes).
 *           M, N multiples of 32; K multiple of 64, <= 16384.
 * The accumulation is exact (i32); it differs from the reference's fp32 simulation by that
 * simulation's own accumulation round-off (~1e-6 relative).                                       */
enum { TQ_ACT_NONE = 0, TQ_ACT_RELU = 1, TQ_ACT_GELU = 2, TQ_ACT_TANH = 3 };
int tq_rowsum_i8(const int8_t* w_idx, int32_t* rowsum, uint64_t N, uint64_t K, tq_stream_t stream);
int tq_linear_i8_fwd(const int8_t* x_idx, const int8_t* w_idx, const int32_t* w_rowsum,
                     const float* bias, void* y,
                     int8_t* y_idx /* optional int8(index - 128) of y for a following integer Linear */,
                     int y_dtype, uint64_t M, uint64_t N, uint64_t K,
                     const float* x_delta, const float* x_zero_float, int x_n_bits, float x_eps,
                     const float* w_delta, uint64_t w_n_params, float w_eps, int activation,
                     const tq_quantizer* q_out, tq_stream_t stream);

/* Activation + output quantizer of tq_linear_i8_fwd as a STAIRCASE TABLE (csrc/tq_stair.hip).  For a per-tensor <= 8-bit
 * q_out,  h(v) = clamp(rne(RN32(act(v)) / scale) + zp, lo, hi) - zp  is a step function of the fp32 pre-activation v with
 * <= 255 steps; RN32(act(v)) is the CORRECTLY ROUNDED fp32 activation (float64 evaluation of nn.GELU()'s erf form, then
 * narrowed), followed by the reference quantizer's own arithmetic (quantizers.py:184-185).  tq_act_stair_build tabulates
 * h on the device over n_bins uniform bins of v (one launch, no host read of the range buffers; rebuild whenever the
 * quantizer's buffers change) into table[tq_act_stair_bytes(n_bins)]: a 16-byte header {1 / bin width, offset, n_bins - 1,
 * ok} and 8 bytes per bin.  ok = 0 when some bin would hold two steps (grid finer than ~n_bins / 75 steps per unit of v):
 * consumers then keep the arithmetic epilogue.  tq_linear_i8_stair_fwd == tq_linear_i8_fwd, except that with a table
 * whose header says ok the epilogue evaluates activation + quantizer by one table read per output (measured at
 * M = 8192: 26.7 -> 21.6 VALU instructions per output over the whole kernel, VALU busy time 13.8 -> 9.3 us);
 * `activation` and `q_out` must be the ones the table was built for.  The table is used by
 * the LDS-tiled kernels (M, N % 64 == 0, K % 128 == 0) when it fits beside the operand stages (n_bins <= 800 for 64 x 64
 * tiles, <= 1664 for 128 x 128); otherwise, and with act_stair == NULL, the call IS tq_linear_i8_fwd.
 * activation: TQ_ACT_NONE / TQ_ACT_RELU / TQ_ACT_GELU.                                                              */
size_t tq_act_stair_bytes(uint32_t n_bins);
int tq_act_stair_build(int activation, const tq_quantizer* q_out, uint32_t n_bins, void* table, size_t table_bytes,
                       tq_stream_t stream);
int tq_linear_i8_stair_fwd(const int8_t* x_idx, const int8_t* w_idx, const int32_t* w_rowsum,
                           const float* bias, void* y, int8_t* y_idx, int y_dtype, uint64_t M, uint64_t N, uint64_t K,
                           const float* x_delta, const float* x_zero_float, int x_n_bits, float x_eps,
                           const float* w_delta, uint64_t w_n_params, float w_eps, int activation,
                           const tq_quantizer* q_out, const void* act_stair, uint32_t stair_bins, tq_stream_t stream);

/* Linear -> (+ residual) -> NoNorm -> quantizers as one launch (MobileBERT bottlenecks / residual tails; reference
 * models/quantized_mobilebert.py:58-72 with :287-304, :330-352 behind hijacker.py:66-116):
 *   residual == NULL:  y = Q_out( Q_dense(lin) * nn_weight + nn_bias )
 *   else:              y = Q_out( Q_sum( Q_dense(lin) + residual ) * nn_weight + nn_bias )
 * lin as in tq_linear_i8_fwd (no activation function); residual fp32 [M, N]; nn_weight / nn_bias fp32 [N], already
 * fake-quantized; every quantizer per-tensor, NULL = identity.  Bit-identical to tq_linear_i8_fwd followed by
 * tq_residual_nonorm_quant_fwd / tq_affine_fake_quant_fwd.                                                        */
int tq_linear_i8_nonorm_fwd(const int8_t* x_idx, const int8_t* w_idx, const int32_t* w_rowsum, const float* bias,
                            const float* residual, const float* nn_weight, const float* nn_bias, void* y,
                            int8_t* y_idx, int y_dtype, uint64_t M, uint64_t N, uint64_t K, const float* x_delta,
                            const float* x_zero_float, int x_n_bits, float x_eps, const float* w_delta,
                            uint64_t w_n_params, float w_eps, const tq_quantizer* q_dense,
                            const tq_quantizer* q_sum, const tq_quantizer* q_out, tq_stream_t stream);
/* Two such Linear -> NoNorm chains that read the SAME input, as one launch (MobileBERT's `bottleneck.input` and
 * `bottleneck.attention`: two QuantizedBottleneckLayer, models/quantized_mobilebert.py:404-417, built at :483-488, both
 * 512 -> 128 from the layer input): weights, row
 * sums, biases, per-row weight scales and the NoNorm affine parameters stacked along N; q_dense[g] / q_out[g] the
 * per-tensor quantizers of group g (all groups or none).  Group g's output is a tensor of its own: y + g * M * N/G
 * ([M, N/G] each; y_idx likewise).  Bit-identical to G tq_linear_i8_nonorm_fwd calls without residual.  G = 3 takes a plain
 * quantized Linear on the same input along (MobileBERT's value Linear, :216-226 called at :507-513) as a chain with the
 * identity affine map and q_dense[2] == q_out[2]: the fixed-range quantizer is idempotent on its own grid, so y / y_idx
 * of that group are exactly the Linear's quantized output and its indices.                                             */
int tq_linear_i8_nonorm_grouped_fwd(const int8_t* x_idx, const int8_t* w_idx, const int32_t* w_rowsum, const float* bias,
                                    const float* nn_weight, const float* nn_bias, void* y, int8_t* y_idx, int y_dtype,
                                    uint64_t M, uint64_t N, uint64_t K, const float* x_delta, const float* x_zero_float,
                                    int x_n_bits, float x_eps, const float* w_delta /* [N] */, float w_eps,
                                    uint64_t n_groups /* 2 or 3 */, const tq_quantizer* const* q_dense,
                                    const tq_quantizer* const* q_out, tq_stream_t stream);

/* MobileBERT feed-forward block as ONE launch (reference models/quantized_mobilebert.py:330-352, hijacker.py:66-116):
 *   y = Q_out( Q_sum( Q_dense( lin2( Q_mid( relu( lin1(x) ) ) ) ) + residual ) * nn_weight + nn_bias )
 * lin1: [N1, K1] with ReLU and the intermediate quantizer q_mid (per-tensor, asymmetric, <= 8 bit), lin2: [N2, N1], then the
 * NoNorm tail of tq_linear_i8_nonorm_fwd.  The [M, N1] intermediate never leaves the CU (its int8 indices are written
 * into LDS in the layout the second GEMM reads).  Built for (K1, N1, N2) = (128, 512, 128), M % 32 == 0; other shapes
 * return TQ_EINVAL.  Bit-identical to tq_linear_i8_fwd followed by tq_linear_i8_nonorm_fwd.                          */
int tq_ffn_i8_nonorm_fwd(const int8_t* x_idx, const float* x_delta, const float* x_zero_float, int x_n_bits, float x_eps,
                         const int8_t* w1_idx, const int32_t* w1_rowsum, const float* bias1, const float* w1_delta,
                         uint64_t w1_n_params, float w1_eps, const tq_quantizer* q_mid, const int8_t* w2_idx,
                         const int32_t* w2_rowsum, const float* bias2, const float* w2_delta, uint64_t w2_n_params,
                         float w2_eps, const float* residual, const float* nn_weight, const float* nn_bias,
                         const tq_quantizer* q_dense, const tq_quantizer* q_sum, const tq_quantizer* q_out, void* y,
                         int8_t* y_idx, int y_dtype, uint64_t M, uint64_t K1, uint64_t N1, uint64_t N2,
                         tq_stream_t stream);

/* A CHAIN of such feed-forward blocks as ONE launch: block s + 1 reads block s's output (as GEMM input and as residual).
 * A MobileBERT layer runs four in a row (reference models/quantized_mobilebert.py:523-529), and NoNorm has no row
 * statistics, so a workgroup takes its 16 token rows through all of them without leaving the CU; only the last block's
 * output is written.  x_idx / x_delta / ... / residual describe the input of block 0 (residual = its fp32 values);
 * `stages` is a HOST array read during the call; every block but the last needs an asymmetric <= 8-bit q_out (its grid is
 * the next block's input grid).  Same shape restriction as tq_ffn_i8_nonorm_fwd, M % 16 == 0.  Bit-identical to n_stages
 * calls of tq_ffn_i8_nonorm_fwd.                                                                                    */
typedef struct tq_ffn_stage {
  const int8_t*  w1_idx;      const int32_t* w1_rowsum;  const float* bias1;   const float* w1_delta;
  uint64_t       w1_n_params; float          w1_eps;
  const tq_quantizer* q_mid;
  const int8_t*  w2_idx;      const int32_t* w2_rowsum;  const float* bias2;   const float* w2_delta;
  uint64_t       w2_n_params; float          w2_eps;
  const float*   nn_weight;   const float*   nn_bias;
  const tq_quantizer *q_dense, *q_sum, *q_out;
} tq_ffn_stage;
int tq_ffn_chain_i8_nonorm_fwd(const int8_t* x_idx, const float* x_delta, const float* x_zero_float, int x_n_bits, float x_eps,
                               const float* residual, const tq_ffn_stage* stages, uint64_t n_stages /* 2..4 */, void* y,
                               int8_t* y_idx, int y_dtype, uint64_t M, uint64_t K1, uint64_t N1, uint64_t N2,
                               tq_stream_t stream);

/* Several quantized Linears that share their input, as ONE launch: the weights (and row sums, biases,
 * per-row weight scales w_delta[N]) of n_groups <= 3 layers are stacked along N; group g owns output
 * columns [g N / n_groups, (g+1) N / n_groups) and has its own per-tensor output quantizer q_out[g]
 * (BERT: query | key | value, models/quantized_bert.py:135-146).  y may be NULL (index-only output:
 * the attention core reads int8 indices).  M % 64 == 0, K % 128 == 0, N / n_groups % 64 == 0.        */
int tq_linear_i8_grouped_fwd(const int8_t* x_idx, const int8_t* w_idx, const int32_t* w_rowsum,
                             const float* bias, void* y, int8_t* y_idx, int y_dtype, uint64_t M,
                             uint64_t N, uint64_t K, const float* x_delta, const float* x_zero_float,
                             int x_n_bits, float x_eps, const float* w_delta, float w_eps,
                             int activation, uint64_t n_groups, const tq_quantizer* const* q_out,
                             tq_stream_t stream);

/* Fixed-range quantized self-attention core on the i8 matrix cores (reference
 * models/quantized_bert.py:135-213: head split, Q K^T, score quantizer, 1/sqrt(d), mask, softmax,
 * probability quantizer, P V, head merge, context quantizer -- 2 batched fp32 GEMMs, 4 permute copies
 * and 6 element-wise sweeps upstream).  q_idx / k_idx / v_idx: int8(index - 128) [B, T, H * head_dim] as
 * emitted by the producing Linears (TQ_IDX_I8_M128 / y_idx), with their per-tensor asymmetric <= 8-bit
 * quantizers q_q / q_k / q_v; q_probs likewise (required); q_scores / q_ctx per-tensor or NULL.
 * mask: additive fp32 [B, T] or NULL.  ctx fp32 [B, T, H * head_dim]; ctx_idx optional int8(index - 128)
 * of ctx (needs an asymmetric <= 8-bit q_ctx).  T a multiple of 64, <= 512; head_dim 64 (BERT) or 32 (MobileBERT).  Both GEMMs are
 * exact integer contractions with zero-point corrections; scores and probabilities never reach HBM. */
int tq_attention_i8_fwd(const int8_t* q_idx, const int8_t* k_idx, const int8_t* v_idx, float* ctx,
                        int8_t* ctx_idx, uint64_t B, uint64_t T, uint64_t H, uint64_t head_dim,
                        uint64_t qkv_row_stride /* elements between tokens of q/k/v; 0 = H*head_dim;
                                                   3*H*head_dim inside a stacked Q|K|V buffer */,
                        const float* mask, float denom, const tq_quantizer* q_q,
                        const tq_quantizer* q_k, const tq_quantizer* q_v, const tq_quantizer* q_scores,
                        const tq_quantizer* q_probs, const tq_quantizer* q_ctx, tq_stream_t stream);
/* The same with a row stride of its own for v_idx (0 = H*head_dim): MobileBERT's query and key Linears share their input
 * (models/quantized_mobilebert.py:214-226 called with the bottlenecked shared input at :507-513) and are one grouped launch whose
 * [B, T, 2 * H * head_dim] index buffer q_idx / k_idx point into, while the value Linear reads the layer input.        */
int tq_attention_i8_strided_fwd(const int8_t* q_idx, const int8_t* k_idx, const int8_t* v_idx, float* ctx,
                                int8_t* ctx_idx, uint64_t B, uint64_t T, uint64_t H, uint64_t head_dim,
                                uint64_t qk_row_stride, uint64_t v_row_stride, const float* mask, float denom,
                                const tq_quantizer* q_q, const tq_quantizer* q_k, const tq_quantizer* q_v,
                                const tq_quantizer* q_scores, const tq_quantizer* q_probs, const tq_quantizer* q_ctx,
                                tq_stream_t stream);

/* Fused attention probabilities with fixed ranges (reference models/quantized_bert.py:153-198):
 *     probs = Q_probs( softmax( Q_scores(scores) / denom + mask, dim=-1 ) )
 * scores / probs fp32 [rows, cols] (rows = B*H*T_query, cols = T_key in {32,64,128,256,512,1024}); mask
 * fp32 [rows / rows_per_mask, cols] additive (BERT: [B, T], rows_per_mask = H * T_query) or NULL;
 * each quantizer per-tensor or NULL.  1 read + 1 write instead of 5 sweeps.                       */
int tq_scores_softmax_quant_fwd(const float* scores, float* probs, uint64_t rows, uint64_t cols,
                                const float* mask, uint64_t rows_per_mask, float denom,
                                const tq_quantizer* q_scores, const tq_quantizer* q_probs,
                                tq_stream_t stream);

/* STE backward of the same op (autograd through quantizers.py:12-19,
 * 184-185, 209): dx = ((g * scale) * mask) / scale with mask = [int_min <= round(x/s)+zp <=
 * int_max]: 3 streams, 6 B/elem bf16.  Non-NULL grad_delta / grad_zero_float (fp32 [n_params], overwritten)
 * also receive d loss / d _delta and d loss / d _zero_float
 * (make_range_trainable, quantizers.py:284-288, 346-349) through deterministic block partials in
 * `workspace` (tq_fake_quant_bwd_workspace_bytes).                                                */
size_t tq_fake_quant_bwd_workspace_bytes(uint64_t n);
/* Per-channel / per-axis quantizers (n_params > 1: `learn_ranges()` on per-channel weights or per-embedding / PEG
 * activations, which the reference differentiates through plain autograd): grad_delta / grad_zero_float are
 * fp32 [n_params]; workspace as below (one deterministic reduction block per parameter and slice).            */
size_t tq_fake_quant_bwd_params_workspace_bytes(uint64_t n, uint64_t n_params, uint64_t inner);
int tq_fake_quant_bwd(const void* x, const void* grad_y, void* grad_x, float* grad_delta,
                      float* grad_zero_float, uint64_t n, int dtype, const tq_quantizer* q,
                      void* workspace, size_t workspace_bytes, tq_stream_t stream);

/* ---- K4/K5: min / max statistics ------------------------------------------------------------
 * Replaces torch.min/torch.max and the transpose+view+min(-1)/max(-1) chains of the range
 * estimators (quantization/range_estimators.py:82-85,114-116,118-130,142-143,153-160,178-207).
 * out_min/out_max: fp32 [n_params].  Two launches: block partials into `workspace`, then a
 * finalize kernel; deterministic (no atomics).                                                */
size_t tq_minmax_workspace_bytes(uint64_t n, uint64_t n_params, uint64_t inner);
int tq_minmax(const void* x, uint64_t n, int dtype, uint64_t n_params, uint64_t inner,
              float* out_min, float* out_max, void* workspace, size_t workspace_bytes,
              tq_stream_t stream);

/* ---- estimator state update -------------------------------------------------------------------
 * CurrentMinMaxEstimator (incl. per-embedding-group fold and the range-sorted permutation,
 * range_estimators.py:87-112), AllMinMaxEstimator (:162-167) and RunningMinMaxEstimator's EMA
 * (:209-214) applied to fresh batch statistics.
 *   new_min/new_max [n]  batch statistics (per embedding dim when n_groups > 0)
 *   cur_min/cur_max [n]  estimator state, updated in place
 *   initialised          0 on the first batch (state is overwritten)
 *   momentum             python float of the reference (double): the kernel narrows
 *                        (1 - momentum) and momentum to fp32 exactly as ATen does
 *   n_groups             0 = none; otherwise n % n_groups == 0 and every dim receives the
 *                        min/max of its group.  order (int64 [n], may be NULL) is
 *                        argsort(ranges): group g holds dims order[g*gs .. (g+1)*gs).          */
int tq_range_update(int mode, const float* new_min, const float* new_max, float* cur_min,
                    float* cur_max, uint64_t n, int initialised, double momentum,
                    uint64_t n_groups, const int64_t* order, tq_stream_t stream);

/* ---- fused calibration step ----------------------------------------------------------------------
 * The estimating branch of QuantizationManager.forward (quantization_manager.py:99-106) for the
 * min/max estimators in ONE call: batch statistics (tq_minmax) -> estimator update
 * (tq_range_update) -> range-to-parameters (tq_set_range_*) [-> quantize, if y != NULL]; the middle
 * two steps are a single launch.  prev_min/prev_max: estimator state before this batch (NULL on the
 * first batch); cur_min/cur_max/delta/zero_float|signed_flag: fresh outputs [n_params].
 * n_params <= 4096; single-rank only (sharded calibration needs the all-reduce between the steps
 * and uses the separate entry points).                                                           */
size_t tq_calibrate_workspace_bytes(uint64_t n, uint64_t n_params, uint64_t inner);
int tq_calibrate_minmax(const void* x, uint64_t n, int dtype, uint64_t n_params, uint64_t inner,
                        int mode, const float* prev_min, const float* prev_max, float* cur_min,
                        float* cur_max, double momentum, uint64_t n_groups, const int64_t* order,
                        int n_bits, int symmetric, float eps, int log_domain, float* delta,
                        float* zero_float, uint8_t* signed_flag, void* y, void* workspace,
                        size_t workspace_bytes, tq_stream_t stream);

/* The same step for ONE range (per-tensor quantizer), as 2 launches instead of 4: the statistics kernel's
 * last-finishing block (ticket in *counter) also applies the estimator rule and writes cur_min / cur_max
 * / delta / zero_float | signed_flag (each [1]); then the quantizer runs if y != NULL.  *counter
 * (device, 4 bytes) must be 0 on entry and is 0 again when the kernel has finished; workspace as
 * tq_calibrate_workspace_bytes(n, 1, 1).  The result does not depend on which block finishes last.
 * With y != NULL and 16-byte aligned x / y the step is ticket-free (round 3): launch 1 leaves one (min, max) pair per
 * block plus a copy of the previous state in the workspace, launch 2 folds the pairs in every block, applies the rule,
 * derives the parameters and quantizes (block 0 stores the new state; in-place state is safe, the live buffers are not
 * read by launch 2): 3.7 + 4.3 us instead of 8.3 + 3.8 us for a [8, 128, 768] tensor; *counter is then not touched.  */
int tq_calibrate_tensor(const void* x, uint64_t n, int dtype, int mode, const float* prev_min,
                        const float* prev_max, float* cur_min, float* cur_max, double momentum,
                        int n_bits, int symmetric, float eps, int log_domain, float* delta,
                        float* zero_float, uint8_t* signed_flag, void* y, void* workspace,
                        size_t workspace_bytes, uint32_t* counter, tq_stream_t stream);

/* Sharded calibration (calibration batches split across ranks, one process per GPU): the fused
 * step split at the exchange.  tq_calibrate_stats writes the LOCAL shard's statistics as stats[2 * n_params] =
 * [-min | max] straight from the statistics kernel (one launch for a single range of <= 512 blocks, statistics +
 * finalize otherwise); the caller all-reduces `stats` IN PLACE with MAX over RCCL (min and max fused into one
 * collective; exact, min/max are associative); tq_calibrate_apply then performs the estimator update +
 * range -> parameters (one launch) and, if y != NULL, the quantizer: 3 launches + 1 collective per
 * calibrating call -- 2 launches for a single range with FRESH output buffers (cur_* not aliasing prev_*) and
 * 16-byte aligned x / y: every block of the quantizer launch re-derives the parameters from `stats`, block 0
 * stores the new state (in-place state keeps the separate update launch: a block might still read prev_*).  Arguments as tq_calibrate_minmax; counter as tq_calibrate_tensor (may be NULL: always
 * the two-launch statistics); workspace tq_calibrate_workspace_bytes(n, n_params, inner).                  */
int tq_calibrate_stats(const void* x, uint64_t n, int dtype, uint64_t n_params, uint64_t inner,
                       float* stats, void* workspace, size_t workspace_bytes, uint32_t* counter,
                       tq_stream_t stream);
int tq_calibrate_apply(const float* stats, const void* x, uint64_t n, int dtype, uint64_t n_params,
                       uint64_t inner, int mode, const float* prev_min, const float* prev_max,
                       float* cur_min, float* cur_max, double momentum, uint64_t n_groups,
                       const int64_t* order, int n_bits, int symmetric, float eps, int log_domain,
                       float* delta, float* zero_float, uint8_t* signed_flag, void* y,
                       tq_stream_t stream);

/* ---- latency-optimised exchange for sharded calibration: P2P "mailbox" MAX all-reduce of <= 8 KB ---------------
 * One process per GPU.  Every rank allocates ONE mailbox (tq_mailbox_alloc: fine-grained device memory owned by the
 * library -- the single exception to "the caller supplies every buffer": it must be mappable into other processes and
 * outlive every launch), publishes its 64-byte IPC handle (e.g. torch.distributed.all_gather_object) and maps the
 * peers' (tq_mailbox_open, hipIpc over xGMI).  tq_mailbox_allreduce_max then replaces the tiny ncclAllReduce(MAX) of
 * the [-min | max] statistics buffer by ONE small kernel: post the vector + a sequence flag in the own mailbox, spin
 * on every peer's flag, fold the peers' vectors with max (csrc/tq_mailbox.hip).  In place on `stats` (fp32 [n],
 * n <= tq_mailbox_max_floats() = 2048); `peer_bases` is a DEVICE array of `world` mailbox pointers (entry `rank` is
 * ignored); `status` (device, 4 bytes, zero-initialised) gets bit 0 set if a peer did not answer within `spin_budget`
 * polls (0 = default: ~10 minutes, the order of c10d's collective timeout), in which case `stats` is NaN -- the
 * kernel cannot hang for ever; callers must read `status` when calibration ends.  hipGraph-capturable (the sequence
 * number lives in the mailbox).  All ranks must issue the same sequence of calls.                                 */
size_t tq_mailbox_bytes(void);
size_t tq_mailbox_max_floats(void);
size_t tq_mailbox_handle_bytes(void);
int tq_mailbox_alloc(void** base, void* ipc_handle_out /* host, tq_mailbox_handle_bytes() */);
int tq_mailbox_open(const void* ipc_handle /* host */, void** peer_base);
int tq_mailbox_close(void* peer_base);
int tq_mailbox_free(void* base);
int tq_mailbox_allreduce_max(float* stats, uint64_t n, void* my_base, void* const* peer_bases,
                             uint32_t world, uint32_t rank, uint32_t* status, uint32_t spin_budget,
                             tq_stream_t stream);

/* tq_calibrate_stats -> tq_mailbox_allreduce_max -> tq_calibrate_apply as ONE call (3-4 launches): the sharded
 * calibrating step with the host cost of the single-GPU fused step.  2 * n_params <= tq_mailbox_max_floats();
 * workspace >= tq_calibrate_workspace_bytes(n, n_params, inner); counter as tq_calibrate_tensor (may be NULL).    */
int tq_calibrate_minmax_mailbox(const void* x, uint64_t n, int dtype, uint64_t n_params, uint64_t inner,
                                int mode, const float* prev_min, const float* prev_max, float* cur_min,
                                float* cur_max, double momentum, uint64_t n_groups, const int64_t* order,
                                int n_bits, int symmetric, float eps, int log_domain, float* delta,
                                float* zero_float, uint8_t* signed_flag, void* y, void* workspace,
                                size_t workspace_bytes, uint32_t* counter, void* my_base,
                                void* const* peer_bases, uint32_t world, uint32_t rank, uint32_t* status,
                                uint32_t spin_budget, tq_stream_t stream);

/* ---- raw-RCCL exchange for sharded calibration (no torch.distributed / c10d in the data path) ---------------------
 * The loop being sharded is pass_data_for_range_estimation (reference utils/utils.py:47-79: every rank runs it on its
 * slice of the calibration batch); the reference has no collective code, these entry points are what its estimators
 * (range_estimators.py:83-169,172-216: the min/max reductions; :248-256: the candidate-loss accumulation) would call
 * between "local statistic" and "update".  One process per GPU, one communicator per process:
 *   tq_comm_load(path)       host: bind librccl at run time (dlopen; NULL = the librccl already mapped into the process,
 *                            else the loader's search path).  Implicit in the other calls.
 *   tq_comm_get_unique_id    host: rank 0 makes the tq_comm_unique_id_bytes() (=128) byte id; the caller ships it to the
 *                            other ranks (e.g. through the torch.distributed rendezvous store).
 *   tq_comm_init             host, collective over all ranks: ncclCommInitRank on the calling thread's current device.
 *   tq_comm_allreduce        in place on `buf` (device), enqueued on `stream`; no host synchronisation, capturable.
 *   tq_comm_broadcast        in place, from `root`.
 * All ranks must issue the same sequence of collectives.  Errors: TQ_EUNSUPPORTED if librccl cannot be bound.      */
enum { TQ_COMM_F32 = 0, TQ_COMM_F64 = 1, TQ_COMM_I32 = 2, TQ_COMM_U8 = 3 };
enum { TQ_COMM_MAX = 0, TQ_COMM_SUM = 1, TQ_COMM_MIN = 2 };
size_t tq_comm_unique_id_bytes(void);
int tq_comm_load(const char* librccl_path /* host, may be NULL */);
int tq_comm_version(void);                     /* ncclGetVersion code of the bound library, 0 if none */
int tq_comm_get_unique_id(void* id_out /* host */);
int tq_comm_init(const void* unique_id /* host */, int rank, int world, void** comm_out);
int tq_comm_destroy(void* comm);
int tq_comm_abort(void* comm);                 /* ncclCommAbort: tear down without waiting for the peers (rejected set-up) */
int tq_comm_rank_world(void* comm, int* rank /* host */, int* world /* host */);
int tq_comm_allreduce(void* comm, void* buf, uint64_t count, int dtype, int op, tq_stream_t stream);
int tq_comm_broadcast(void* comm, void* buf, uint64_t count, int dtype, int root, tq_stream_t stream);

/* tq_calibrate_stats -> ncclAllReduce(MAX) on [-min | max] -> tq_calibrate_apply as ONE call: the sharded calibrating
 * step of a quantizer with the host cost of the single-GPU fused step.  Arguments as tq_calibrate_minmax; counter as
 * tq_calibrate_tensor (may be NULL); workspace >= tq_calibrate_workspace_bytes(n, n_params, inner).                 */
int tq_calibrate_minmax_rccl(const void* x, uint64_t n, int dtype, uint64_t n_params, uint64_t inner,
                             int mode, const float* prev_min, const float* prev_max, float* cur_min,
                             float* cur_max, double momentum, uint64_t n_groups, const int64_t* order,
                             int n_bits, int symmetric, float eps, int log_domain, float* delta,
                             float* zero_float, uint8_t* signed_flag, void* y, void* workspace,
                             size_t workspace_bytes, uint32_t* counter, void* comm, tq_stream_t stream);

/* PEG phase 1 (range_estimators.py:68-80): ranges = max - min per embedding dim; on later
 * batches the reference stores 0.1*r + 0.9*r of the NEW ranges (quirk q4).                    */
int tq_axis_ranges(const float* new_min, const float* new_max, float* ranges, uint64_t n,
                   int first, tq_stream_t stream);

/* ---- range -> quantizer parameters ---------------------------------------------------------
 * AsymmetricUniformQuantizer.set_quant_range (quantizers.py:234-282) and
 * SymmetricUniformQuantizer.set_quant_range (:334-344), same fp32 operation order.            */
int tq_set_range_asym(const float* x_min, const float* x_max, uint64_t n, int n_bits, float eps,
                      int log_domain, float* delta, float* zero_float, tq_stream_t stream);
int tq_set_range_sym(const float* x_min, const float* x_max, uint64_t n, int n_bits, float eps,
                     int log_domain, float* delta, uint8_t* signed_flag, tq_stream_t stream);

/* ---- K7/K8: MSE range search: loss of many candidate quantizers in one pass ---------------------
 * Replaces the per-candidate deepcopy + set_quant_range + fake-quant + (x-y)^2 + sum loop of
 * MSE_Estimator (range_estimators.py:248-256, 287-294, 356-420) and apply_mse_init
 * (adaround/adaround.py:160-178).
 *   x          [rows, row_len] contiguous (rows = 1 for per-tensor, = channels for per_channel)
 *   cand       fp32 [C,4] = (scale, zero_point, int_min, int_max) per candidate
 *   loss       fp64 [rows, C]; this batch's sum of squared errors is ADDED to it
 *   workspace  tq_mse_workspace_bytes(...)                                                      */
size_t tq_mse_workspace_bytes(uint64_t rows, uint64_t row_len, uint64_t n_cand);
int tq_mse_candidates(const void* x, uint64_t rows, uint64_t row_len, int dtype,
                      const float* cand, uint64_t n_cand, double* loss, void* workspace,
                      size_t workspace_bytes, tq_stream_t stream);

/* Extension: true per-embedding-group search.  The reference's MSE estimator
 * ignores axis / n_groups, so "PEG + MSE" degenerates to a per-tensor search; this entry point
 * evaluates the candidates per group of embedding dimensions of x [n_tokens, d] in place
 * (group g = columns [g*d/n_groups, (g+1)*d/n_groups)), loss fp64 [n_groups, C] accumulated.
 * Oracle: the reference's MSE_Estimator(per_channel=True) on the transposed [n_groups, -1] view.
 * Workspace: tq_mse_workspace_bytes(n_groups, n_tokens * d / n_groups, n_cand).                  */
int tq_mse_candidates_grouped(const void* x, uint64_t n_tokens, uint64_t d, uint64_t n_groups,
                              int dtype, const float* cand, uint64_t n_cand, double* loss,
                              void* workspace, size_t workspace_bytes, tq_stream_t stream);

/* The same losses in the REFERENCE'S OWN fp32 summation order.  MSE_Estimator.loss_fx
 * (range_estimators.py:248-256) returns torch.sum(torch.sum(err.view(len(data), -1), dim=1)) as an fp32
 * value produced by ATen's CPU cascade-sum kernel; scipy's bounded Brent search (golden section, :296-327,
 * :422-470) and the numpy argmin of the grid searches (:370, :405) consume that value, so bit-equal
 * thresholds need the same rounding sequence, not a more accurate sum.  This entry point reproduces it
 * (32 accumulator columns x 4 cascade levels per row = one half-wave per row span, see
 * csrc/tq_mse_ordered.hip; restated and pinned against torch.sum in oracle/aten_sum.py):
 *   x          [rows, row_len] contiguous, rows = len(data) of the reference's view
 *   reduce_rows 1: loss[c] / loss_f32[c] <- the second torch.sum over the row sums (per_channel_loss=False)
 *              0: loss[r, c] / loss_f32[r, c] <- the row sums themselves (per_channel_loss=True)
 *   loss       fp64, the fp32 result is ADDED to it (the reference accumulates batches in a float64 numpy
 *              array, :366, :399); may be NULL
 *   loss_f32   fp32, overwritten with the value itself (golden section); may be NULL
 * Single-threaded ATen order (the reference's CPU run is thread-count dependent only for >= 32768 rows). */
size_t tq_mse_ordered_workspace_bytes(uint64_t rows, uint64_t row_len, uint64_t n_cand);
int tq_mse_candidates_ordered(const void* x, uint64_t rows, uint64_t row_len, int dtype,
                              const float* cand, uint64_t n_cand, int reduce_rows, double* loss,
                              float* loss_f32, void* workspace, size_t workspace_bytes,
                              tq_stream_t stream);
/* K9: CrossEntropyEstimator.loss_fx (range_estimators.py:498-502) for all candidates:
 * loss[c] += -sum softmax(x,dim=1) * log_softmax(Q_c(x),dim=1), x fp32 [rows, cols].          */
int tq_xent_candidates(const float* x, uint64_t rows, uint64_t cols, const float* cand,
                       uint64_t n_cand, double* loss, tq_stream_t stream);

/* argmin over candidates + threshold lookup, on the device (replaces the numpy argmin and
 * host->device copies at range_estimators.py:370-376, 405-420):
 * cur_min[r] = thr_min[argmin_c loss[r,c]], cur_max likewise; ties -> lowest c (np.argmin).   */
int tq_argmin_select(const double* loss, uint64_t rows, uint64_t n_cand, const float* thr_min,
                     const float* thr_max, float* cur_min, float* cur_max, int64_t* best,
                     tq_stream_t stream);

/* Order statistics: out[row, j] = the element of rank ranks[j] (0-based, ascending; NaN sorts last) of each row of x
 * viewed as [rows, n], 1 <= m <= TQ_OSTAT_MAX_RANKS ranks per call, selected EXACTLY by a radix select -- no sort.
 * Replaces the reference's host-side np.percentile (quantization/range_estimators.py:121-140: `to_numpy(x)` +
 * np.percentile(data, (p, 100 - p)) for CurrentMinMaxEstimator's percentile option): method 'linear' interpolates
 * between the two order statistics at floor / ceil of (n - 1) p / 100; the caller does that interpolation with numpy's
 * own arithmetic (quantization/range_estimators.py `_percentile_rows`).  `ranks` is a HOST array (passed on by value);
 * `out` fp32 device [rows, m]; n < 2^32.  No host synchronisation. */
enum { TQ_OSTAT_MAX_RANKS = 4 };
size_t tq_order_stats_workspace_bytes(uint64_t rows, uint32_t m);
int tq_order_stats(const void* x, uint64_t rows, uint64_t n, int dtype, const uint64_t* ranks /* host */, uint32_t m,
                   float* out, void* workspace, size_t workspace_bytes, tq_stream_t stream);

/* ---- K10/K11/K13: AdaRound --------------------------------------------------------------------
 * K10: AdaRoundQuantizer.to_integer_forward + dequantise (adaround/quantizer.py:46-90,
 *      quantizers.py:209): w_q = scale * (clamp(floor(w/s) + r (+zp), lo, hi) - zp),
 *      r = h(alpha) if soft else [alpha >= 0].  w, alpha, w_q fp32 [n].                         */
int tq_adaround_fwd(const float* w, const float* alpha, float* w_q, uint64_t n,
                    const tq_quantizer* q, int mode, int soft, float temperature,
                    tq_stream_t stream);
/* backward of K10 alone (what autograd produces for alpha): grad_alpha = grad_wq * scale *
 * h'(alpha) * [lo <= floor(w/s) + h(alpha) (+zp) <= hi].  Used when an external torch optimizer
 * owns alpha; the fused path below is what apply_adaround_to_layer runs.                        */
int tq_adaround_bwd(const float* w, const float* alpha, const float* grad_wq, float* grad_alpha,
                    uint64_t n, const tq_quantizer* q, int mode, float temperature,
                    tq_stream_t stream);
/* alpha initialisation such that h(alpha) = frac(w/s) (adaround/quantizer.py:57-71).          */
int tq_adaround_init_alpha(const float* w, float* alpha, uint64_t n, const tq_quantizer* q,
                           int mode, float temperature, tq_stream_t stream);
/* K11: fused backward of K10 + rounding regulariser + Adam step on alpha
 *      (adaround/utils.py:159-162, adaround/adaround.py:98-99,260; torch.optim.Adam defaults):
 *      g = grad_wq * scale * h'(alpha) * [lo <= floor(w/s)+h(+zp) <= hi]
 *          + reg_weight * d/dalpha (1 - |2h-1|^beta)           (reg_weight = 0 during warm-up)
 *      then m,v,alpha updated in place; `step` is the 1-based Adam step count.
 *      grad_alpha_out (optional) receives g.                                                      */
int tq_adaround_bwd_adam(const float* w, const float* grad_wq, float* alpha, float* exp_avg,
                         float* exp_avg_sq, float* grad_alpha_out, uint64_t n,
                         const tq_quantizer* q, int mode, float temperature, float reg_weight,
                         float beta, float lr, float adam_b1, float adam_b2, float adam_eps,
                         int step, tq_stream_t stream);
/* K11 with its per-iteration scalars in device memory: sched[4] = {reg_weight, beta, 1 - b1^t, sqrt(1 - b2^t)}
 * (fp32).  Nothing in the argument list changes between iterations: the AdaRound loop body (adaround/adaround.py:
 * 236-262) can be recorded once as a hipGraph and replayed.                                                    */
int tq_adaround_bwd_adam_sched(const float* w, const float* grad_wq, float* alpha, float* exp_avg,
                               float* exp_avg_sq, uint64_t n, const tq_quantizer* q, int mode, float temperature,
                               const float* sched, float lr, float adam_b1, float adam_b2, float adam_eps,
                               tq_stream_t stream);
/* regulariser value: out[0] += weight * sum(1 - |2h(alpha)-1|^beta)   (fp64 accumulate).        */
int tq_adaround_reg(const float* alpha, uint64_t n, int mode, float temperature, float beta,
                    float weight, double* out, void* workspace, size_t workspace_bytes,
                    tq_stream_t stream);
size_t tq_reduce_workspace_bytes(uint64_t n);
/* K13: reconstruction loss (adaround/utils.py:150): mse(pred,tgt,'none').sum(1).mean() for
 *      fp32 [d0, d1, rest]; out[0] (fp64) is overwritten.                                         */
int tq_recon_loss(const float* pred, const float* tgt, uint64_t d0, uint64_t d1, uint64_t rest,
                  double* out, void* workspace, size_t workspace_bytes, tq_stream_t stream);

/* ---- FP64 (`--double`, reference main.py:227-231: modules cast to float64) --------------------------------
 * The quantizer path with every operation in IEEE double: fake-quant forward / STE backward (quantizers.py:
 * 172-211, 291-349), batch min / max (range_estimators.py:82-85, 114-130), estimator state update (:87-112,
 * 162-167, 183-193, 209-214), PEG ranges (:68-80), range -> parameters (quantizers.py:234-282, 334-344) and the
 * MSE candidate loss (range_estimators.py:248-256; the candidate table stays fp32 like the fp32 tensors the
 * reference builds from python-float thresholds).  Same (n_params, inner) layout convention as tq_quantizer.   */
typedef struct tq_quantizer_f64 {
  const double*  delta;
  const double*  zero_float;   /* NULL for symmetric */
  const uint8_t* signed_flag;
  int32_t        n_bits;       /* 1..52 */
  int32_t        symmetric;
  int32_t        log_domain;
  int32_t        reserved;
  double         eps;
  uint64_t       n_params;
  uint64_t       inner;
} tq_quantizer_f64;
int tq_fake_quant_fwd_f64(const double* x, double* y /* or NULL */, double* idx /* or NULL */, uint64_t n,
                          const tq_quantizer_f64* q, tq_stream_t stream);
size_t tq_fake_quant_bwd_f64_workspace_bytes(uint64_t n, uint64_t n_params, uint64_t inner);
int tq_fake_quant_bwd_f64(const double* x, const double* grad_y, double* grad_x,
                          double* g_delta /* [n_params] or NULL */, double* g_zero_float /* [n_params] or NULL */,
                          uint64_t n, const tq_quantizer_f64* q, void* workspace, size_t workspace_bytes,
                          tq_stream_t stream);
size_t tq_minmax_f64_workspace_bytes(uint64_t n, uint64_t n_params, uint64_t inner);
int tq_minmax_f64(const double* x, uint64_t n, uint64_t n_params, uint64_t inner, double* out_min, double* out_max,
                  void* workspace, size_t workspace_bytes, tq_stream_t stream);
int tq_range_update_f64(int mode, const double* new_min, const double* new_max, double* cur_min, double* cur_max,
                        uint64_t n, int initialised, double momentum, uint64_t n_groups, const int64_t* order,
                        tq_stream_t stream);
int tq_axis_ranges_f64(const double* new_min, const double* new_max, double* ranges, uint64_t n, int first,
                       tq_stream_t stream);
int tq_set_range_asym_f64(const double* x_min, const double* x_max, uint64_t n, int n_bits, double eps, int log_domain,
                          double* delta, double* zero_float, tq_stream_t stream);
int tq_set_range_sym_f64(const double* x_min, const double* x_max, uint64_t n, int n_bits, double eps, int log_domain,
                         double* delta, uint8_t* signed_flag, tq_stream_t stream);
int tq_mse_candidates_f64(const double* x, uint64_t rows, uint64_t row_len, const float* cand /* [n_cand, 4] */,
                          uint64_t n_cand, int reduce_rows, double* loss /* += [rows | 1, n_cand] */,
                          tq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TQ_HIP_H */
