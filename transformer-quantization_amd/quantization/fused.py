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
from quantization import provenance
from quantization.base_quantized_classes import FP32Acts
from quantization.quantization_manager import QuantizationManager, Qstates, _GLOBAL_FWD_HOOKS, _GLOBAL_FWD_PRE_HOOKS


def _fixed_per_tensor(enabled, mgr):
    """-> ('off' | 'no' | 7-tuple): disabled quantizer, not fusable, or kernel arguments."""
    if not enabled or isinstance(mgr, FP32Acts):
        return 'off'
    if not isinstance(mgr, QuantizationManager) or mgr.state != Qstates.fix_ranges:
        return 'no'
    q = mgr.quantizer
    if mgr._forward_hooks or mgr._forward_pre_hooks or q._forward_hooks or q._forward_pre_hooks:
        return 'no'                      # an observer on a stage the fused launch would skip: layered route
    if not q.is_initialized or q._delta.numel() != 1:
        return 'no'
    return (q._delta, q._zero_float, getattr(q, '_signed', None), q.n_bits, q.symmetric,
            q.scale_domain == 'log', q.eps)


def _hooked(*modules):
    """forward (pre-)hooks on modules a fused launch would not call -- leaf modules AND the containers whose __call__
    a merged launch bypasses (the harness models pass those: QResidualBlock, QFFN, the Sequential around an intermediate
    Linear, ...) -- or global module hooks (torch.nn.modules.module.register_module_forward_hook), which every bypassed
    __call__ would have fired."""
    if _GLOBAL_FWD_HOOKS or _GLOBAL_FWD_PRE_HOOKS:
        return True
    return any(m is not None and isinstance(m, torch.nn.Module) and (m._forward_hooks or m._forward_pre_hooks)
               for m in modules)


hooked = _hooked      # public name for the harness models (containers they bypass when they take a merged launch)


def residual_layernorm_quant(dense, res_quantizer, layer_norm, x, residual, _gemm=None):
    """dense: QuantLinear, res_quantizer: QuantizedActivation, layer_norm: QuantLayerNorm or MobileBERT's
    QuantNoNorm (tq_residual_nonorm_quant_fwd).  Equivalent to ``layer_norm(res_quantizer(dense(x) + residual))``.
    (_gemm: the pre-quantizer output of `dense`, already computed by quantized_bert_ffn from int8 indices.)"""
    if _gemm is not None:
        x = _gemm
    q1 = _fixed_per_tensor(dense._quant_a and dense.activation_function is None, dense.activation_quantizer)
    q2 = _fixed_per_tensor(getattr(res_quantizer, '_quant_a', False),
                           getattr(res_quantizer, 'activation_quantizer', res_quantizer))   # FP32Acts: site switched off
    q3 = _fixed_per_tensor(layer_norm._quant_a and layer_norm.activation_function is None,
                           layer_norm.activation_quantizer)
    from quantization.autoquant_utils import QuantNoNorm
    is_nonorm = isinstance(layer_norm, QuantNoNorm)
    fusable = ('no' not in (q1, q2, q3) and dense.activation_function is None
               and layer_norm.activation_function is None and _hip.on_device(x) and x.dtype != torch.float64
               and not (torch.is_grad_enabled() and (x.requires_grad or residual.requires_grad))
               and (is_nonorm or len(layer_norm.normalized_shape) == 1)
               and dense.activation_save_target is None and layer_norm.activation_save_target is None
               and not _hooked(dense, res_quantizer, layer_norm))
    if not fusable:
        if _gemm is not None:
            raise RuntimeError('residual_layernorm_quant: pre-computed GEMM handed to a tail that is not fusable')
        return layer_norm(res_quantizer(dense(x) + residual))
    if is_nonorm and _gemm is None:
        y = _linear_nonorm_i8(dense, layer_norm, x, residual, q1, q2, q3)      # GEMM + whole tail as ONE integer launch
        if y is not None:
            return y
    gemm = _gemm
    if gemm is None and options.int8_active() and hasattr(dense, '_int8_forward'):
        gemm = dense._int8_forward(x, with_output_quantizer=False)     # exact integer GEMM (MFMA i8)
    if gemm is None:
        w, b = dense.get_params()
        gemm = dense.run_forward(x, w, b)                   # hipBLASLt through torch (fp32 simulation)
    if is_nonorm:
        # one weight quantizer serves weight and bias, in this order (upstream quirk, autoquant_utils.QuantNoNorm)
        ln_w, ln_b = layer_norm.quantized_params()          # cached with fixed ranges in inference
    else:
        ln_w, ln_b = layer_norm.get_params()                # fake-quantized (cached in eval) affine
    arg = lambda q: None if q == 'off' else q
    oq = layer_norm.activation_quantizer.quantizer if q3 != 'off' else None
    want_idx = (options.int8_active() and oq is not None and not oq.symmetric and oq.n_bits <= 8
                and gemm.dtype == torch.float32)
    out = _hip.backend().residual_layernorm_quant(gemm, residual, arg(q1), arg(q2), ln_w, ln_b,
                                                  None if is_nonorm else layer_norm.eps, arg(q3), want_idx=want_idx)
    y = out[0] if want_idx else out
    if oq is not None:
        provenance.tag(y, oq, out[1] if want_idx else None)
    return y


_LN_VECTORS = (16, 32, 64, 96, 128, 192, 256, 384, 512, 768)      # 16-byte vectors per row the tail kernels are built for


def embeddings_layernorm_quant(word_emb, type_emb, pos_emb, sum1, sum2, layer_norm, input_ids, type_ids, pos_ids):
    """BERT's embedding block (reference models/quantized_bert.py:75-111):

        layer_norm( sum2( sum1( word_emb(input_ids) + type_emb(type_ids) ) + pos_emb(pos_ids) ) )

    with three QuantEmbeddings (no output quantizer), two QuantizedActivations and a QuantLayerNorm, as ONE launch
    (tq_embeddings_layernorm_quant_fwd: the 13 launches of the layered modules read and write [B, T, d] nine times) when
    every range involved is fixed and per-tensor.  Returns None otherwise: the caller runs the layered modules."""
    from quantization.autoquant_utils import QuantEmbedding, QuantLayerNorm
    be = _hip.backend()
    embs = (word_emb, type_emb, pos_emb)
    if (not hasattr(be, 'embeddings_layernorm_quant') or not _hip.on_device(input_ids) or input_ids.dim() != 2
            or input_ids.dtype != torch.int64 or type(layer_norm) is not QuantLayerNorm
            or layer_norm.activation_function is not None or len(layer_norm.normalized_shape) != 1
            or layer_norm.activation_save_target is not None or layer_norm.training
            or _needs_autograd(word_emb, type_emb, pos_emb, layer_norm) or _hooked(*embs, sum1, sum2, layer_norm)):
        return None
    for e in embs:
        if (type(e) is not QuantEmbedding or e.training or e.max_norm is not None or e.activation_save_target is not None
                or not isinstance(e.activation_quantizer, FP32Acts) or _hooked(e._modules.get('weight_quantizer'))):
            return None
    q1 = _fixed_per_tensor(getattr(sum1, '_quant_a', False), getattr(sum1, 'activation_quantizer', sum1))
    q2 = _fixed_per_tensor(getattr(sum2, '_quant_a', False), getattr(sum2, 'activation_quantizer', sum2))
    q3 = _fixed_per_tensor(layer_norm._quant_a, layer_norm.activation_quantizer)
    if 'no' in (q1, q2, q3):
        return None
    tables = [e.get_params()[0] for e in embs]                  # fake-quantized (cached in eval mode) fp32 tables
    d = tables[0].shape[-1]
    if (any(t.dtype != torch.float32 or t.shape[-1] != d or not _hip.on_device(t) for t in tables) or d % 4
            or d // 4 not in _LN_VECTORS):
        return None
    B, T = input_ids.shape
    if pos_ids.shape[-1] != T or type_ids.shape != input_ids.shape:
        return None
    pos_full = pos_ids.expand(B, T) if pos_ids.shape[0] != B else pos_ids
    ln_w, ln_b = layer_norm.get_params()
    arg = lambda q: None if q == 'off' else q
    oq = layer_norm.activation_quantizer.quantizer if q3 != 'off' else None
    want_idx = options.int8_active() and oq is not None and not oq.symmetric and oq.n_bits <= 8
    out = be.embeddings_layernorm_quant(tables[0], input_ids, tables[1], type_ids, tables[2], pos_full, arg(q1), arg(q2),
                                        ln_w, ln_b, layer_norm.eps, arg(q3), want_idx=want_idx)
    y = (out[0] if want_idx else out).view(B, T, d)
    if oq is not None:
        provenance.tag(y, oq, out[1].view(B, T, d) if want_idx else None)
    return y


def quantized_bert_ffn(intermediate, dense, res_quantizer, layer_norm, x, residual):
    """BERT feed-forward block (reference models/quantized_bert.py:252-280 behind hijacker.py:66-116):

        layer_norm(res_quantizer(dense(intermediate(x)) + residual))

    with `intermediate` a QuantLinear + GELU whose 8-bit output ONLY feeds `dense`.  With options.INT8_LINEAR and fixed
    per-tensor ranges everywhere, the intermediate Linear runs index-only (tq_linear_i8_fwd with y = NULL: the
    [tokens, 3072] fp32 activation -- 12.6 MB per layer at B = 8, 4/5 of that kernel's HBM writes -- is never stored),
    `dense` consumes the int8 indices, and the residual + LayerNorm tail follows as one kernel.  Same integer
    contractions and element arithmetic as the separate calls: bit-identical result.  Anything else: the layered
    modules / the tail helper."""
    def separate():
        return residual_layernorm_quant(dense, res_quantizer, layer_norm, intermediate(x), residual)

    if (not options.int8_active() or not hasattr(intermediate, '_int8_plan') or not hasattr(dense, '_int8_plan_from')
            or not _hip.on_device(x) or x.dtype != torch.float32 or dense.activation_function is not None
            or _needs_autograd(intermediate, dense, layer_norm, x, residual)
            or _hooked(intermediate, dense, res_quantizer, layer_norm)):
        return separate()
    from quantization.autoquant_utils import QuantNoNorm
    q1 = _fixed_per_tensor(dense._quant_a, dense.activation_quantizer)
    q2 = _fixed_per_tensor(getattr(res_quantizer, '_quant_a', False), getattr(res_quantizer, 'activation_quantizer', res_quantizer))
    q3 = _fixed_per_tensor(layer_norm._quant_a and layer_norm.activation_function is None, layer_norm.activation_quantizer)
    if ('no' in (q1, q2, q3) or isinstance(layer_norm, QuantNoNorm) or layer_norm.activation_function is not None
            or len(layer_norm.normalized_shape) != 1 or dense.activation_save_target is not None
            or layer_norm.activation_save_target is not None or intermediate.activation_save_target is not None
            or residual.dtype != torch.float32):
        return separate()                                  # (the same conditions the tail helper checks)
    plan1 = intermediate._int8_plan(x, with_output_quantizer=True)
    if plan1 is None or plan1[2] is None or plan1[2][4] or plan1[2][5] or plan1[2][3] > 8:
        return separate()                                  # intermediate quantizer: asymmetric, linear domain, <= 8 bit
    mid_q = intermediate.activation_quantizer.quantizer
    M = x.numel() // intermediate.in_features
    plan2 = dense._int8_plan_from(mid_q, M, with_output_quantizer=False)
    if plan2 is None or not dense._int8_weights()[2]:
        return separate()
    mid_idx = intermediate._int8_compute(x, plan1, index_only=True)
    if mid_idx is None:
        return separate()
    gemm = dense._int8_compute(None, plan2, x_idx=mid_idx)
    if gemm is None:
        return separate()
    return residual_layernorm_quant(dense, res_quantizer, layer_norm, None, residual, _gemm=gemm)


def _needs_autograd(*modules_and_tensors):
    """True when a result must carry a grad_fn: grad mode is on and an input tensor or a parameter of one of the given
    modules requires grad.  The fused integer kernels are inference-only; checking the input alone would silently drop
    the gradients of Linear / NoNorm weights behind a frozen input (frozen embeddings, first layer)."""
    if not torch.is_grad_enabled():
        return False
    for m in modules_and_tensors:
        if m is None:
            continue
        if torch.is_tensor(m):
            if m.requires_grad:
                return True
        elif any(p.requires_grad for p in m.parameters()):
            return True
    return False


def _linear_nonorm_i8(dense, layer_norm, x, residual, q1, q2, q3):
    """Integer Linear -> (+ residual -> Q_sum) -> NoNorm -> Q_out in one launch (tq_linear_i8_nonorm_fwd), or None when
    the integer path does not apply to `dense` / `x` (then the caller runs GEMM and tail separately).  Bit-identical to
    that two-launch form: same integer contraction, same element arithmetic."""
    if not options.int8_active() or not hasattr(dense, '_int8_plan') or (residual is not None and (
            residual.dtype != torch.float32 or not _hip.on_device(residual))) or _needs_autograd(dense, layer_norm, x, residual):
        return None
    plan = dense._int8_plan(x, with_output_quantizer=False)
    if plan is None or plan[1] != _hip.ACT_NONE:
        return None
    ops = dense._int8_operands(x, plan)
    if ops is None:
        return None
    from quantization.autoquant_utils import INT8_STATS
    arg = lambda q: None if q == 'off' else q
    ln_w, ln_b = layer_norm.quantized_params()
    oq = layer_norm.activation_quantizer.quantizer if q3 != 'off' else None
    want_idx = oq is not None and not oq.symmetric and oq.n_bits <= 8
    INT8_STATS['kernel_calls'] += 1
    out = _hip.backend().linear_i8_nonorm(ops[0], ops[1], ops[2], ops[3], residual, ln_w, ln_b, ops[4], ops[5], ops[6],
                                          arg(q1), arg(q2), arg(q3), torch.float32, want_idx=want_idx)
    y = out[0] if want_idx else out
    if oq is not None:
        provenance.tag(y, oq, out[1] if want_idx else None)
    return y


def linear_nonorm_quant(dense, layer_norm, x):
    """MobileBERT bottleneck: ``layer_norm(dense(x))`` with a QuantLinear and a QuantNoNorm, fixed per-tensor ranges:
    one integer launch when options.INT8_LINEAR applies, the layered modules otherwise."""
    from quantization.autoquant_utils import QuantNoNorm
    if (isinstance(layer_norm, QuantNoNorm) and _hip.on_device(x) and x.dtype == torch.float32
            and dense.activation_function is None and layer_norm.activation_function is None
            and dense.activation_save_target is None and layer_norm.activation_save_target is None
            and not _needs_autograd(dense, layer_norm, x) and not _hooked(dense, layer_norm)):
        q1 = _fixed_per_tensor(dense._quant_a, dense.activation_quantizer)
        q3 = _fixed_per_tensor(layer_norm._quant_a, layer_norm.activation_quantizer)
        if 'no' not in (q1, q3):
            y = _linear_nonorm_i8(dense, layer_norm, x, None, q1, 'off', q3)
            if y is not None:
                return y
    return layer_norm(dense(x))


def linear_nonorm_quant_pair(dense_a, layer_norm_a, dense_b, layer_norm_b, x, value=None):
    """MobileBERT's two input bottlenecks (reference models/quantized_mobilebert.py:404-417, built at :483-488):

        layer_norm_a(dense_a(x)), layer_norm_b(dense_b(x))        [, value(x)]

    -- two QuantLinear -> QuantNoNorm chains reading the SAME tensor -- as ONE integer launch
    (tq_linear_i8_nonorm_grouped_fwd) when options.INT8_LINEAR applies and every range involved is fixed and per-tensor;
    each result is a contiguous tensor of its own, tagged with its quantizer and int8 indices.  `value`: the attention's
    value QuantLinear, which reads x too (:216-226 called at :507-513): it rides along as a third group -- a chain with the
    identity affine map whose two quantizers are the Linear's own output quantizer (idempotent on its grid: the group's
    output IS value(x), bit for bit) -- when it has the bottlenecks' width and an asymmetric <= 8-bit fixed output
    quantizer; a third result is returned then.  None when the pair itself is not eligible (the caller runs the chains
    separately).  Bit-identical to the separate launches."""
    from quantization.autoquant_utils import QuantLinear, QuantNoNorm, INT8_STATS, _fixed_per_tensor_manager
    be = _hip.backend()
    pairs = ((dense_a, layer_norm_a), (dense_b, layer_norm_b))
    if (not options.int8_active() or not hasattr(be, 'linear_i8_nonorm_grouped') or not _hip.on_device(x)
            or x.dtype != torch.float32 or _needs_autograd(dense_a, layer_norm_a, dense_b, layer_norm_b, x)
            or _hooked(dense_a, layer_norm_a, dense_b, layer_norm_b)):
        return None
    K = x.shape[-1]
    M = x.numel() // K
    N = dense_a.out_features
    if M % 64 or K % 128 or K > 16384 or N % 64 or dense_b.out_features != N:
        return None
    q_dense, q_out, plans = [], [], []
    for d, ln in pairs:
        if (type(d) is not QuantLinear or type(ln) is not QuantNoNorm or d.training or ln.training or d.in_features != K
                or d.activation_function is not None or ln.activation_function is not None
                or d.activation_save_target is not None or ln.activation_save_target is not None
                or not hasattr(d, '_int8_plan') or _hooked(getattr(d, 'weight_quantizer', None))):
            return None
        q1 = _fixed_per_tensor(d._quant_a, d.activation_quantizer)
        q3 = _fixed_per_tensor(ln._quant_a, ln.activation_quantizer)
        if 'no' in (q1, q3):
            return None
        plan = d._int8_plan(x, with_output_quantizer=False)
        if plan is None or plan[1] != _hip.ACT_NONE:
            return None
        q_dense.append(q1)
        q_out.append(q3)
        plans.append(plan)
    if ((q_dense[0] == 'off') != (q_dense[1] == 'off') or (q_out[0] == 'off') != (q_out[1] == 'off')
            or dense_a.weight_quantizer.quantizer.eps != dense_b.weight_quantizer.quantizer.eps):
        return None
    oqs = [ln.activation_quantizer.quantizer if q != 'off' else None for (_, ln), q in zip(pairs, q_out)]
    layers = [dense_a, dense_b]
    # the value Linear as a third chain: needs the same structure of quantizers as the bottlenecks (both present)
    if (value is not None and q_dense[0] != 'off' and q_out[0] != 'off' and type(value) is QuantLinear and not value.training
            and value.in_features == K and value.out_features == N and value.activation_function is None
            and value.activation_save_target is None and hasattr(value, '_int8_plan') and value._quant_a
            and _fixed_per_tensor_manager(value.activation_quantizer) and not _needs_autograd(value)
            and not _hooked(value, getattr(value, 'weight_quantizer', None))
            and value.weight_quantizer.quantizer.eps == dense_a.weight_quantizer.quantizer.eps):
        vq = value.activation_quantizer.quantizer
        vplan = value._int8_plan(x, with_output_quantizer=False)
        if (vplan is not None and vplan[1] == _hip.ACT_NONE and not vq.symmetric and vq.n_bits <= 8
                and vq.scale_domain == 'linear'):
            q7 = (vq._delta, vq._zero_float, None, vq.n_bits, False, False, vq.eps)
            layers.append(value)
            q_dense.append(q7)
            q_out.append(q7)
            oqs.append(vq)
    G = len(layers)
    want_idx = all(oq is not None and not oq.symmetric and oq.n_bits <= 8 for oq in oqs)
    packed = _stacked_qkv(tuple(layers))
    if packed is None and G == 3:
        layers, q_dense, q_out, oqs, G = layers[:2], q_dense[:2], q_out[:2], oqs[:2], 2
        packed = _stacked_qkv(tuple(layers))
    if packed is None:
        return None
    w_idx, rowsum, bias, scales = packed
    ops = dense_a._int8_operands(x, plans[0])              # the shared input's indices and quantizer
    if ops is None:
        return None
    affine = [ln.quantized_params() for _, ln in pairs]    # cached with fixed ranges in inference
    # quantized pairs: fresh tensors per (parameter versions, range state) from quantized_params' own cache; with
    # _quant_w = False they ARE the raw Parameters, which an optimizer step / load_state_dict rewrites in place behind the
    # same pointers: versions and CACHE_EPOCH (hipGraph replays of a training step) are part of the key
    key = tuple((w.data_ptr(), w._version, b.data_ptr(), b._version) for w, b in affine) + (G, options.CACHE_EPOCH)
    hit = getattr(layer_norm_a, '_stacked_affine_cache', None)
    if hit is None or hit[0] != key:
        ws = [w.detach().float().reshape(-1) for w, _ in affine]
        bs = [b.detach().float().reshape(-1) for _, b in affine]
        if G == 3:                                         # identity affine map of the value group
            ws.append(torch.ones(N, dtype=torch.float32, device=x.device))
            bs.append(torch.zeros(N, dtype=torch.float32, device=x.device))
        hit = (key, torch.cat(ws).contiguous(), torch.cat(bs).contiguous(), affine)   # (keeps the parts alive)
        layer_norm_a._stacked_affine_cache = hit
    INT8_STATS['kernel_calls'] += G
    out = be.linear_i8_nonorm_grouped(ops[0], w_idx, rowsum, bias, hit[1], hit[2], ops[4], scales, ops[6],
                                      None if q_dense[0] == 'off' else q_dense, None if q_out[0] == 'off' else q_out,
                                      torch.float32, want_idx=want_idx, n_groups=G)
    ys, idxs = out if want_idx else (out, [None] * G)
    res = []
    for y, idx, oq in zip(ys, idxs, oqs):
        y = y.view(x.shape[:-1] + (N,))
        if oq is not None:
            provenance.tag(y, oq, None if idx is None else idx.view(y.shape))
        res.append(y)
    return res


def quantized_ffn(intermediate, dense, res_quantizer, layer_norm, x):
    """MobileBERT feed-forward block (reference models/quantized_mobilebert.py:330-352):

        layer_norm(res_quantizer(dense(intermediate(x)) + x))

    with `intermediate` a QuantLinear + ReLU, `dense` a QuantLinear, `layer_norm` a QuantNoNorm -- as ONE integer launch
    (tq_ffn_i8_nonorm_fwd: the [tokens, 512] intermediate stays in LDS) when options.INT8_LINEAR applies to both Linears,
    every range involved is fixed and per-tensor and the shape is one the kernel is built for; otherwise the two
    Linears run on their own (integer or layered) and the tail through residual_layernorm_quant."""
    def separate():
        return residual_layernorm_quant(dense, res_quantizer, layer_norm, intermediate(x), x)

    from quantization.autoquant_utils import INT8_STATS, QuantNoNorm
    be = _hip.backend()
    if (not options.int8_active() or not hasattr(be, 'ffn_i8_nonorm') or not isinstance(layer_norm, QuantNoNorm)
            or not hasattr(intermediate, '_int8_plan') or not hasattr(dense, '_int8_weight_side_ok')
            or not _hip.on_device(x) or x.dtype != torch.float32 or _needs_autograd(intermediate, dense, layer_norm, x)
            or dense.activation_function is not None or layer_norm.activation_function is not None
            or layer_norm.activation_save_target is not None or not dense._int8_weight_side_ok()
            or _hooked(intermediate, dense, res_quantizer, layer_norm)):
        return separate()
    K1, N1, N2 = intermediate.in_features, intermediate.out_features, dense.out_features
    if (K1, N1, N2) not in be.FFN_SHAPES or dense.in_features != N1 or (x.numel() // K1) % 32:
        return separate()
    plan = intermediate._int8_plan(x, with_output_quantizer=True)
    q1 = _fixed_per_tensor(dense._quant_a, dense.activation_quantizer)
    q2 = _fixed_per_tensor(getattr(res_quantizer, '_quant_a', False), getattr(res_quantizer, 'activation_quantizer', res_quantizer))
    q3 = _fixed_per_tensor(layer_norm._quant_a, layer_norm.activation_quantizer)
    if plan is None or plan[1] != _hip.ACT_RELU or plan[2] is None or 'no' in (q1, q2, q3):
        return separate()
    q_mid = plan[2]
    if q_mid[4] or q_mid[5] or q_mid[3] > 8:                 # intermediate quantizer: asymmetric, linear, <= 8 bit
        return separate()
    ops = intermediate._int8_operands(x, plan)
    if ops is None:
        return separate()
    w2_idx, rs2, w2_signed = dense._int8_weights()
    if not w2_signed:
        return separate()
    arg = lambda q: None if q == 'off' else q
    ln_w, ln_b = layer_norm.quantized_params()
    oq = layer_norm.activation_quantizer.quantizer if q3 != 'off' else None
    want_idx = oq is not None and not oq.symmetric and oq.n_bits <= 8
    wq2 = dense.weight_quantizer.quantizer
    bias2 = None if dense.bias is None else dense.bias.detach()
    INT8_STATS['kernel_calls'] += 2                          # two Linears' worth of integer GEMM
    out = be.ffn_i8_nonorm(ops[0], ops[4], ops[1], ops[2], ops[3], ops[5], ops[6], q_mid, w2_idx, rs2, bias2,
                           wq2._delta.reshape(-1), wq2.eps, x, ln_w, ln_b, arg(q1), arg(q2), arg(q3), torch.float32,
                           want_idx=want_idx)
    y = out[0] if want_idx else out
    if oq is not None:
        provenance.tag(y, oq, out[1] if want_idx else None)
    return y


def quantized_ffn_chain(blocks, x):
    """Consecutive MobileBERT feed-forward blocks, each the input of the next (reference
    models/quantized_mobilebert.py:523-529: three FFN layers, then intermediate + output), as ONE integer launch
    (tq_ffn_chain_i8_nonorm_fwd: a workgroup takes its 16 token rows through all blocks; only the last output is
    written).  blocks: [(intermediate, dense, res_quantizer, layer_norm), ...] as for `quantized_ffn`.  Same
    preconditions as `quantized_ffn` for every block, and every block's output quantizer must be asymmetric, linear,
    <= 8 bit (its grid is the next block's input grid).  None when not eligible (the caller runs the blocks one by one);
    bit-identical to that."""
    from quantization.autoquant_utils import INT8_STATS, QuantNoNorm
    be = _hip.backend()
    if (not options.int8_active() or not hasattr(be, 'ffn_chain_i8_nonorm') or not 2 <= len(blocks) <= 4
            or not _hip.on_device(x) or x.dtype != torch.float32):
        return None
    K1 = x.shape[-1]
    M = x.numel() // K1
    if M % 32:
        return None
    arg = lambda q: None if q == 'off' else q
    stages, src, x_ops, oq = [], None, None, None
    for k, (intermediate, dense, res_quantizer, layer_norm) in enumerate(blocks):
        if (not isinstance(layer_norm, QuantNoNorm) or not hasattr(intermediate, '_int8_plan')
                or not hasattr(dense, '_int8_weight_side_ok') or _needs_autograd(intermediate, dense, layer_norm, x)
                or dense.activation_function is not None or layer_norm.activation_function is not None
                or layer_norm.activation_save_target is not None or dense.activation_save_target is not None
                or intermediate.activation_save_target is not None or not dense._int8_weight_side_ok()
                or _hooked(intermediate, dense, res_quantizer, layer_norm)
                or (intermediate.in_features, intermediate.out_features, dense.out_features) not in be.FFN_SHAPES
                or dense.in_features != intermediate.out_features):
            return None
        plan = (intermediate._int8_plan(x, with_output_quantizer=True) if k == 0
                else intermediate._int8_plan_from(src, M, with_output_quantizer=True))
        q1 = _fixed_per_tensor(dense._quant_a, dense.activation_quantizer)
        q2 = _fixed_per_tensor(getattr(res_quantizer, '_quant_a', False), getattr(res_quantizer, 'activation_quantizer', res_quantizer))
        q3 = _fixed_per_tensor(layer_norm._quant_a, layer_norm.activation_quantizer)
        if plan is None or plan[1] != _hip.ACT_RELU or plan[2] is None or 'no' in (q1, q2, q3) or q3 == 'off':
            return None
        q_mid = plan[2]
        oq = layer_norm.activation_quantizer.quantizer
        if q_mid[4] or q_mid[5] or q_mid[3] > 8 or oq.symmetric or oq.n_bits > 8 or oq.scale_domain != 'linear':
            return None
        if k == 0:
            x_ops = intermediate._int8_operands(x, plan)
            ops = x_ops
        else:
            ops = intermediate._int8_operands(None, plan, x_idx=x_ops[0])     # (x_idx unused: weight side only)
        if ops is None:
            return None
        w2_idx, rs2, w2_signed = dense._int8_weights()
        if not w2_signed:
            return None
        ln_w, ln_b = layer_norm.quantized_params()
        wq2 = dense.weight_quantizer.quantizer
        stages.append(dict(w1_idx=ops[1], w1_rowsum=ops[2], bias1=ops[3], w1_delta=ops[5], w1_eps=ops[6], q_mid=q_mid,
                           w2_idx=w2_idx, w2_rowsum=rs2, bias2=None if dense.bias is None else dense.bias.detach(),
                           w2_delta=wq2._delta.reshape(-1), w2_eps=wq2.eps, nn_w=ln_w, nn_b=ln_b,
                           q_dense=arg(q1), q_sum=arg(q2), q_out=q3))
        src = oq
    INT8_STATS['kernel_calls'] += 2 * len(blocks)
    y, idx = be.ffn_chain_i8_nonorm(x_ops[0], x_ops[4], x, stages, torch.float32, want_idx=True)
    provenance.tag(y, oq, idx)
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
    if ('no' in (q1, q2) or not _hip.on_device(scores) or scores.dtype != torch.float32 or scores.dim() != 4
            or not ok_mask or Tk not in (32, 64, 128, 256, 512, 1024)
            or (torch.is_grad_enabled() and scores.requires_grad) or _hooked(scores_quantizer, probs_quantizer)):
        s = scores_quantizer(scores) / denom
        if mask is not None:
            s = s + mask
        return probs_quantizer(torch.softmax(s, dim=-1))
    arg = lambda q: None if q == 'off' else q
    m = None if mask is None else mask.reshape(mask.shape[0], Tk).float().contiguous()
    return _hip.backend().scores_softmax_quant(scores, m, scores.shape[1] * scores.shape[2], denom,
                                               arg(q1), arg(q2))


def _int8_source(t):
    """(int8 indices, 7-tuple) of a tensor produced by a fixed per-tensor asymmetric <= 8-bit quantizer
    that emitted its indices (provenance records of QuantizationManager / the integer Linear)."""
    q, idx = provenance.of(t) or (None, None)
    if (q is None or idx is None or idx.shape != t.shape or q.symmetric or q.n_bits > 8
            or q.scale_domain != 'linear' or q._delta.numel() != 1):
        return None
    return idx, (q._delta, q._zero_float, None, q.n_bits, False, False, q.eps)


def quantized_attention(query, key, value, mask, num_heads, scores_quantizer, probs_quantizer,
                        context_quantizer):
    """Attention core of a quantized BERT layer (reference models/quantized_bert.py:135-213) on the
    OUTPUTS of the quantized query / key / value Linears, each [B, T, H * d]:

        ctx = context_quantizer(merge_heads(probs_quantizer(softmax(
                  scores_quantizer(Q K^T) / sqrt(d) + mask)) V))

    Runs as one integer kernel (tq_attention_i8_fwd) when options.INT8_LINEAR is on, the three inputs
    carry their int8 grid indices, every quantizer involved is fixed, per-tensor (asymmetric <= 8 bit
    for Q, K, V and the probabilities), T a multiple of 64 up to 512 and d in (32, 64).  Returns None otherwise: the
    caller then runs the layered modules."""
    if (not options.int8_active() or query.dim() != 3 or not _hip.on_device(query)
            or _hooked(scores_quantizer, probs_quantizer, context_quantizer)):
        return None
    if torch.is_grad_enabled() and (query.requires_grad or key.requires_grad or value.requires_grad):
        return None
    B, T, D = query.shape
    if D % num_heads or D // num_heads not in (32, 64) or T % 64 or T > 512:
        return None
    srcs = [_int8_source(t) for t in (query, key, value)]
    qs = _fixed_per_tensor(scores_quantizer._quant_a, scores_quantizer.activation_quantizer)
    qp = _fixed_per_tensor(probs_quantizer._quant_a, probs_quantizer.activation_quantizer)
    qc = _fixed_per_tensor(context_quantizer._quant_a, context_quantizer.activation_quantizer)
    if None in srcs or 'no' in (qs, qp, qc) or qp == 'off':
        return None
    if qp[4] or qp[5] or qp[3] > 8:                       # probabilities: asymmetric, linear, <= 8 bit
        return None
    if mask is not None:
        if not (mask.dim() == 4 and mask.shape[0] == B and mask.shape[1] == 1 and mask.shape[2] == 1
                and mask.shape[3] == T):
            return None
        mask = mask.reshape(B, T).float().contiguous()
    arg = lambda q: None if q == 'off' else q
    cq = context_quantizer.activation_quantizer.quantizer if qc != 'off' else None
    want_idx = cq is not None and not cq.symmetric and cq.n_bits <= 8
    out = _hip.backend().attention_i8(srcs[0][0], srcs[1][0], srcs[2][0], num_heads, mask, float(D // num_heads) ** 0.5,
                                      srcs[0][1], srcs[1][1], srcs[2][1], arg(qs), qp, arg(qc), want_idx=want_idx)
    ctx = out[0] if want_idx else out
    if cq is not None:
        provenance.tag(ctx, cq, out[1] if want_idx else None)
    return ctx


def _stacked_qkv(layers):
    """int8 weights / row sums / biases / per-row weight scales of several QuantLinears stacked along the
    output dimension, cached on the first layer until any weight or weight range changes."""
    key = tuple((l.weight.data_ptr(), l.weight._version, l.weight_quantizer.quantizer.range_state_key(),
                 None if l.bias is None else (l.bias.data_ptr(), l.bias._version)) for l in layers)
    cache = getattr(layers[0], '_stacked_i8_cache', None)
    if cache is not None and cache[0] == key:
        return cache[1]
    parts = [l._int8_weights() for l in layers]              # (w_idx, rowsum, signed) per layer
    if not all(p[2] for p in parts):
        return None
    dev = layers[0].weight.device
    w_idx = torch.cat([p[0] for p in parts], dim=0).contiguous()
    rowsum = torch.cat([p[1] for p in parts], dim=0).contiguous()
    if all(l.bias is None for l in layers):
        bias = None
    else:
        bias = torch.cat([l.bias.detach().float() if l.bias is not None else
                          torch.zeros(l.out_features, device=dev) for l in layers]).contiguous()
    scales = torch.cat([l.weight_quantizer.quantizer._delta.detach().reshape(-1).float().expand(l.out_features)
                        if l.weight_quantizer.quantizer._delta.numel() == 1
                        else l.weight_quantizer.quantizer._delta.detach().reshape(-1).float() for l in layers]).contiguous()
    packed = (w_idx, rowsum, bias, scales)
    layers[0]._stacked_i8_cache = (key, packed)
    return packed


def quantized_self_attention(x, query, key, value, mask, num_heads, scores_quantizer, probs_quantizer,
                             context_quantizer, value_out=None):
    """Self-attention of a quantized BERT / MobileBERT layer from the inputs of its query / key / value Linears: Linears
    that share their input run as ONE grouped integer GEMM that only emits int8 indices (tq_linear_i8_grouped_fwd:
    no fp32 output is written), which the integer attention core consumes in place (column blocks of the stacked
    buffers).  `x`: the one layer input (BERT: one launch for Q | K | V), or a (query_in, key_in, value_in) tuple
    (MobileBERT, reference models/quantized_mobilebert.py:214-226 called at :507-513: query and key read the bottlenecked
    shared input -> Q | K in one launch, V in another).  Same preconditions as `quantized_attention` plus: the three
    Linears are plain eval-mode QuantLinears without activation function whose weight and output quantizers are fixed,
    and the inputs carry their int8 indices.  value_out: value(value_in) already computed by another launch
    (linear_nonorm_quant_pair), tagged with the value Linear's output quantizer and its indices -- the value Linear is not
    launched then.  Returns None when any of that does not hold (run the layered modules then)."""
    from quantization.autoquant_utils import QuantLinear, _fixed_per_tensor_manager
    layers = (query, key, value)
    xs = tuple(x) if isinstance(x, (tuple, list)) else (x, x, x)
    if (not options.int8_active() or len(xs) != 3
            or any(t.dim() != 3 or not _hip.on_device(t) or t.dtype != torch.float32 for t in xs)
            or _hooked(query, key, value, scores_quantizer, probs_quantizer, context_quantizer,
                       *(getattr(l, 'weight_quantizer', None) for l in layers))):
        return None
    if torch.is_grad_enabled() and (any(t.requires_grad for t in xs) or any(l.weight.requires_grad for l in layers)):
        return None
    B, T, _ = xs[0].shape
    D = query.out_features
    if (any(t.shape[:2] != (B, T) for t in xs) or D % num_heads or D // num_heads not in (32, 64) or T % 64 or T > 512
            or (B * T) % 64 or D % 64):
        return None
    outs = []
    for l, t in zip(layers, xs):
        K = t.shape[-1]
        if (type(l) is not QuantLinear or l.training or not l._quant_w or not l._quant_a
                or l.activation_function is not None or l.activation_save_target is not None
                or l.in_features != K or l.out_features != D or K % 128 or K > 16384
                or not _fixed_per_tensor_manager(l.activation_quantizer)
                or not isinstance(l.weight_quantizer, QuantizationManager)      # e.g. replaced by FP32Acts
                or l.weight_quantizer.state != Qstates.fix_ranges):
            return None
        wq, oq = l.weight_quantizer.quantizer, l.activation_quantizer.quantizer
        if (not wq.symmetric or wq.n_bits > 8 or wq.scale_domain != 'linear' or wq._delta.numel() not in (1, D)
                or oq.symmetric or oq.n_bits > 8 or oq.scale_domain != 'linear'
                or wq.eps != query.weight_quantizer.quantizer.eps):     # one eps argument serves the stacked weights
            return None
        outs.append((oq._delta, oq._zero_float, None, oq.n_bits, False, False, oq.eps))
    qs = _fixed_per_tensor(scores_quantizer._quant_a, scores_quantizer.activation_quantizer)
    qp = _fixed_per_tensor(probs_quantizer._quant_a, probs_quantizer.activation_quantizer)
    qc = _fixed_per_tensor(context_quantizer._quant_a, context_quantizer.activation_quantizer)
    if 'no' in (qs, qp, qc) or qp == 'off' or qp[4] or qp[5] or qp[3] > 8:
        return None
    if mask is not None:
        if not (mask.dim() == 4 and mask.shape[0] == B and mask.shape[1] == 1 and mask.shape[2] == 1
                and mask.shape[3] == T):
            return None
        mask = mask.reshape(B, T).float().contiguous()
    # consecutive Linears reading the SAME tensor object form one launch: (Q, K, V) | (Q, K), (V) | (Q), (K), (V)
    groups = [[0]]
    for i in (1, 2):
        if xs[i] is xs[groups[-1][0]]:
            groups[-1].append(i)
        else:
            groups.append([i])
    v_src = None
    if value_out is not None:
        rec = provenance.of(value_out)
        v_src = _int8_source(value_out)
        if (rec is None or rec[0] is not value.activation_quantizer.quantizer or v_src is None
                or value_out.shape != (B, T, D) or not value_out.is_contiguous()):
            return None
        groups = [g for g in ([i for i in grp if i != 2] for grp in groups) if g]
    srcs = {}
    for g in groups:
        src = _int8_source(xs[g[0]])
        if src is None:
            return None
        srcs[g[0]] = src
    be = _hip.backend()
    cols = [None, None, None]
    packs = [_stacked_qkv(tuple(layers[i] for i in g)) for g in groups]
    if any(p is None for p in packs):
        return None
    from quantization.autoquant_utils import INT8_STATS
    for g, packed in zip(groups, packs):
        INT8_STATS['kernel_calls'] += len(g)                  # counted in Linears, like the other fused launches
        w_idx, rowsum, bias, scales = packed
        x_idx, xq = srcs[g[0]]
        _, buf = be.linear_i8_grouped(x_idx, w_idx, rowsum, bias, (xq[0], xq[1], xq[3], xq[6]), scales,
                                      layers[g[0]].weight_quantizer.quantizer.eps, 0, [outs[i] for i in g], want_y=False,
                                      want_idx=True)
        for j, i in enumerate(g):
            cols[i] = buf[..., j * D:(j + 1) * D]
    if v_src is not None:
        cols[2] = v_src[0]
    arg = lambda q: None if q == 'off' else q
    cq = context_quantizer.activation_quantizer.quantizer if qc != 'off' else None
    want_idx = cq is not None and not cq.symmetric and cq.n_bits <= 8
    out = be.attention_i8(cols[0], cols[1], cols[2], num_heads, mask, float(D // num_heads) ** 0.5,
                          outs[0], outs[1], outs[2], arg(qs), qp, arg(qc), want_idx=want_idx)
    ctx = out[0] if want_idx else out
    if cq is not None:
        provenance.tag(ctx, cq, out[1] if want_idx else None)
    return ctx
