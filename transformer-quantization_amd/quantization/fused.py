"""Fused fixed-range layer tails -- new, no counterpart module upstream.

``residual_layernorm_quant`` runs the tail of BertSelfOutput / BertOutput
(reference models/quantized_bert.py:238-248, 264-280)

    Q_ln( LayerNorm( Q_res( Q_dense(dense_out) + residual ) ) )

as ONE kernel (``tq_residual_layernorm_quant_fwd``: 2 reads + 1 write of [B*T, d]) when every
quantizer involved has a fixed per-tensor range; otherwise it runs the layered modules (the same HIP
kernels, one launch per stage), so calibration / QAT / per-embedding configurations keep their exact
semantics.  ``scores_softmax_quant`` does the same for the attention probabilities
(``tq_scores_softmax_quant_fwd``: quantizer -> 1/sqrt(d) -> mask -> softmax -> quantizer, 1 read + 1 write).
"""
import torch

from quantization import _hip
from quantization import options
from quantization.base_quantized_classes import FP32Acts
from quantization.quantization_manager import QuantizationManager, Qstates


def _fixed_per_tensor(enabled, mgr):
    """-> ('off' | 'no' | 7-tuple): disabled quantizer, not fusable, or kernel arguments."""
    if not enabled or isinstance(mgr, FP32Acts):
        return 'off'
    if not isinstance(mgr, QuantizationManager) or mgr.state != Qstates.fix_ranges:
        return 'no'
    q = mgr.quantizer
    if not q.is_initialized or q._delta.numel() != 1:
        return 'no'
    return (q._delta, q._zero_float, getattr(q, '_signed', None), q.n_bits, q.symmetric,
            q.scale_domain == 'log', q.eps)


def residual_layernorm_quant(dense, res_quantizer, layer_norm, x, residual):
    """dense: QuantLinear, res_quantizer: QuantizedActivation, layer_norm: QuantLayerNorm.
    Equivalent to ``layer_norm(res_quantizer(dense(x) + residual))``."""
    q1 = _fixed_per_tensor(dense._quant_a and dense.activation_function is None, dense.activation_quantizer)
    q2 = _fixed_per_tensor(res_quantizer._quant_a, res_quantizer.activation_quantizer)
    q3 = _fixed_per_tensor(layer_norm._quant_a and layer_norm.activation_function is None,
                           layer_norm.activation_quantizer)
    fusable = ('no' not in (q1, q2, q3) and dense.activation_function is None
               and layer_norm.activation_function is None and x.is_cuda
               and not (torch.is_grad_enabled() and (x.requires_grad or residual.requires_grad))
               and len(layer_norm.normalized_shape) == 1
               and dense.activation_save_target is None and layer_norm.activation_save_target is None)
    if not fusable:
        return layer_norm(res_quantizer(dense(x) + residual))
    gemm = None
    if options.INT8_LINEAR and hasattr(dense, '_int8_forward'):
        gemm = dense._int8_forward(x, with_output_quantizer=False)     # exact integer GEMM (MFMA i8)
    if gemm is None:
        w, b = dense.get_params()
        gemm = dense.run_forward(x, w, b)                   # hipBLASLt through torch (fp32 simulation)
    ln_w, ln_b = layer_norm.get_params()                    # fake-quantized (cached in eval) affine
    arg = lambda q: None if q == 'off' else q
    oq = layer_norm.activation_quantizer.quantizer if q3 != 'off' else None
    want_idx = (options.INT8_LINEAR and oq is not None and not oq.symmetric and oq.n_bits <= 8
                and gemm.dtype == torch.float32)
    out = _hip.backend().residual_layernorm_quant(gemm, residual, arg(q1), arg(q2), ln_w, ln_b,
                                                  layer_norm.eps, arg(q3), want_idx=want_idx)
    y = out[0] if want_idx else out
    if oq is not None:
        y._tq_quantizer = oq
        if want_idx:
            y._tq_idx = out[1]
    return y


def scores_softmax_quant(scores_quantizer, probs_quantizer, scores, mask, denom):
    """Equivalent to ``probs_quantizer(softmax(scores_quantizer(scores) / denom + mask, dim=-1))``
    (reference models/quantized_bert.py:153-198) as one kernel when both quantizers are fixed and
    per-tensor.  scores: fp32 [B, H, Tq, Tk]; mask: additive, broadcastable [B, 1, 1, Tk] or None."""
    q1 = _fixed_per_tensor(scores_quantizer._quant_a, scores_quantizer.activation_quantizer)
    q2 = _fixed_per_tensor(probs_quantizer._quant_a, probs_quantizer.activation_quantizer)
    Tk = scores.shape[-1]
    ok_mask = mask is None or (mask.dim() == 4 and mask.shape[1] == 1 and mask.shape[2] == 1
                               and mask.shape[0] == scores.shape[0] and mask.shape[3] == Tk)
    if ('no' in (q1, q2) or not scores.is_cuda or scores.dtype != torch.float32 or scores.dim() != 4
            or not ok_mask or Tk not in (32, 64, 128, 256, 512, 1024)
            or (torch.is_grad_enabled() and scores.requires_grad)):
        s = scores_quantizer(scores) / denom
        if mask is not None:
            s = s + mask
        return probs_quantizer(torch.softmax(s, dim=-1))
    arg = lambda q: None if q == 'off' else q
    m = None if mask is None else mask.reshape(mask.shape[0], Tk).float().contiguous()
    return _hip.backend().scores_softmax_quant(scores, m, scores.shape[1] * scores.shape[2], denom,
                                               arg(q1), arg(q2))
