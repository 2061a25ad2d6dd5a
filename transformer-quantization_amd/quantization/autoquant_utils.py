"""Quantized drop-ins for nn.Linear / nn.LayerNorm / nn.Embedding and the recursive
``quantize_model`` rewriter (counterpart of the reference's quantization/autoquant_utils.py).

``run_forward`` stays the per-op hook of the reference (:20-21, :59-66, :76-85); it is also where a
fused GEMM + fake-quant epilogue would plug in.
"""
import copy
import warnings

import torch
from torch import nn
from torch.nn import functional as F
from torch.nn.modules.pooling import _AdaptiveAvgPoolNd, _AvgPoolNd

from quantization import _hip
from quantization import options
from quantization import provenance
from quantization.base_quantized_classes import FP32Acts, QuantizedActivation, QuantizedModule
from quantization.hijacker import QuantizationHijacker, activations_list
from quantization.quantization_manager import QuantizationManager, _GLOBAL_FWD_HOOKS, _GLOBAL_FWD_PRE_HOOKS


# The integer path is switched on with quantization.options.INT8_LINEAR (see there).
# how often the MFMA Linear ran (plain / under autograd) and how often an unsigned weight grid sent a layer back to
# the layered path
INT8_STATS = {'kernel_calls': 0, 'autograd_calls': 0, 'unsigned_weight_fallbacks': 0}
_ACT_CODES = {type(None): _hip.ACT_NONE, nn.ReLU: _hip.ACT_RELU, nn.GELU: _hip.ACT_GELU, nn.Tanh: _hip.ACT_TANH}


def _hooked(*modules):
    """Does any of these modules carry forward (pre-)hooks?  The integer / fused routes evaluate whole module chains in
    one launch without calling the modules, so a hook on any of them (or a global module hook) would silently stop
    firing."""
    if _GLOBAL_FWD_HOOKS or _GLOBAL_FWD_PRE_HOOKS:
        return True
    return any(m is not None and isinstance(m, nn.Module) and (m._forward_hooks or m._forward_pre_hooks) for m in modules)


def _fixed_per_tensor_manager(mgr):
    from quantization.quantization_manager import Qstates
    return (isinstance(mgr, QuantizationManager) and mgr.state == Qstates.fix_ranges
            and mgr.quantizer.is_initialized and mgr.quantizer._delta.numel() == 1)


def prequantize_weights(model):
    """Fill the eval-mode parameter cache of every quantized layer below `model` whose weight range is FIXED with ONE
    multi-tensor launch per 40 tensors (`tq_fake_quant_multi_fwd`) instead of one launch per layer on its first eval
    forward.  The reference quantizes and caches the weights layer by layer (hijacker.py:52-64 `get_params`: 102 weight
    tensors in a BERT-base); this is the same computation on the same inputs -- `cached_params` ends up bit-identical
    -- batched over independent sites.  Layers that do not qualify (training mode, caching off, a cache already
    present, weights not quantized, ranges not fixed / learnable / not initialised, rows that are not whole 16-byte
    vectors, float64) are left to the lazy path.  -> number of layers served."""
    from quantization.quantization_manager import Qstates
    from quantization.quantizers import AsymmetricUniformQuantizer, SymmetricUniformQuantizer, param_layout
    be = _hip.backend()
    if not hasattr(be, 'fake_quant_multi'):
        return 0
    groups = {}                     # (device, dtype) -> [(module, weight, bias, item)]
    for m in model.modules():
        if not isinstance(m, QuantizationHijacker) or m.training or not m._caching or m.cached_params or not m._quant_w:
            continue
        mgr = m._modules.get('weight_quantizer')
        if not isinstance(mgr, QuantizationManager) or mgr.state != Qstates.fix_ranges:
            continue
        if mgr._forward_hooks or mgr._forward_pre_hooks:
            continue
        q = mgr.quantizer
        if type(q) not in (AsymmetricUniformQuantizer, SymmetricUniformQuantizer) or not q.is_initialized:
            continue
        if q._forward_hooks or q._forward_pre_hooks:
            continue
        weight, bias = m.get_weight_bias()
        delta = q._delta
        if (not weight.is_cuda or weight.dtype not in (torch.float32, torch.bfloat16, torch.float16)
                or delta.requires_grad or delta.device != weight.device or not weight.is_contiguous()
                or weight.data_ptr() % 16):
            continue
        try:
            n_params, inner = param_layout(weight, delta.numel(), q.axis, q.per_channel, tuple(delta.shape))
        except (ValueError, RuntimeError):
            continue                # the lazy path raises the reference's error
        vec = 4 if weight.dtype == torch.float32 else 8
        if n_params > 1 and inner % vec:
            continue
        item = (weight.detach(), delta, None if q.symmetric else q._zero_float, getattr(q, '_signed', None), q.n_bits,
                q.symmetric, q.scale_domain == 'log', q.eps, n_params, inner)
        groups.setdefault((weight.device, weight.dtype), []).append((m, bias, item))
    served = 0
    for (device, _), entries in groups.items():
        with torch.cuda.device(device):
            ys = be.fake_quant_multi([e[2] for e in entries])
        for (m, bias, _), y in zip(entries, ys):
            # exactly what get_params caches: detached fp32 copies resident in HBM
            m.cached_params = (y.detach().to(torch.float32), None if bias is None else bias.detach().to(torch.float32))
            served += 1
    return served


def precalibrate_weights(model):
    """Golden-section weight ranges of every layer below `model` in ONE lock-step search
    (`range_estimators.golden_section_lockstep`) instead of one scipy search -- with a host round trip per loss
    evaluation -- per layer on its first calibrating forward.  The README's standard recipe (reference README.md:149-157:
    `--weight-quant-method MSE --weight-opt-method golden_section`) runs 102 such searches of ~35 evaluations each on a
    BERT-base; the weights do not depend on the data, so the searches are independent and advance together: ~35 rounds of
    102 queued launches and one device->host copy each.  Every estimator keeps the thresholds and a memo of the tensor they
    were found for; the calibrating forward that follows (`pass_data_for_range_estimation`) finds them there and
    launches nothing for the search.  Same scipy calls on the same fp32 loss values: `_delta` is bit-identical
    (tests/test_bert_e2e.py::test_bert_base_readme_recipe_*).  Layers that do not qualify (other estimators, per-channel
    ranges, weights not quantized, not in an estimating state, sharded calibration) keep the lazy path.
    -> {'searches', 'rounds', 'evaluations'}"""
    from quantization.range_estimators import golden_section_lockstep, lockstep_eligible
    jobs, devices = [], set()
    for m in model.modules():
        if not isinstance(m, QuantizationHijacker) or not m._quant_w:
            continue
        mgr = m._modules.get('weight_quantizer')
        if not isinstance(mgr, QuantizationManager) or not mgr._estimating() or _hooked(mgr, mgr.range_estimator):
            continue
        weight, _ = m.get_weight_bias()
        if torch.is_tensor(weight) and lockstep_eligible(mgr.range_estimator, weight):
            jobs.append((mgr.range_estimator, weight))
            devices.add(weight.device)
    if len(jobs) < 2 or len(devices) != 1:
        return {'searches': 0, 'rounds': 0, 'evaluations': 0}
    with torch.cuda.device(next(iter(devices))) if next(iter(devices)).type == 'cuda' else _null():
        return golden_section_lockstep(jobs)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def int8_stair_status(model):
    """{QuantLinear name: {n_bins: table accepted by its builder?}} for every integer Linear below `model` that has built a
    GELU staircase table so far.  Reads the tables' headers (a host synchronisation): diagnostics, not the data path."""
    report = {}
    for name, m in model.named_modules():
        tables = getattr(m, '_int8_stair', None)
        if isinstance(m, QuantLinear) and tables:
            report[name] = {n_bins: bool(entry[1][0][:16].view(torch.float32)[3].item() == 1.0)
                            for n_bins, entry in tables.items() if torch.is_tensor(entry[1][0])}
    return report


class QuantLinear(QuantizationHijacker, nn.Linear):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._int8_cache = None
        self._int8_signed = None
        self._int8_stair = None

    def run_forward(self, x, weight, bias, offsets=None):
        return F.linear(x.contiguous(), weight.contiguous(), bias=bias)

    def forward(self, x, offsets=None):
        if options.int8_active() and (options.INT8_LINEAR is True or not self.training):
            amgr = self._modules.get('activation_quantizer')
            calibrating = self._quant_a and type(amgr) is QuantizationManager and amgr._estimating()
            y = self._int8_calibrating_forward(x) if calibrating else self._int8_forward(x)
            if y is not None:
                return y
        return super().forward(x, offsets)

    def _int8_calibrating_forward(self, x):
        """options.INT8_CALIBRATION: this layer's output quantizer is still ESTIMATING its range, but the input lies on
        the grid its producer just set (provenance: the estimating quantizer tags what it returns) and the weight grid
        is known -- so the GEMM is the same exact integer contraction as in a fixed-range forward (the fp32 simulation
        of the layered route approximates this very value), with bias and activation function in the epilogue and the
        un-quantized fp32 result handed to the estimator as usual.  None: not applicable, the layered modules run."""
        amgr = self._modules.get('activation_quantizer')
        if (not options.INT8_CALIBRATION or self.training or not self._quant_a or type(amgr) is not QuantizationManager
                or not amgr._estimating() or torch.is_grad_enabled()):
            return None
        # small GEMMs stay with torch's fp32 GEMM: the integer path costs one more launch (the input's indices) and ~60 us
        # more host time per layer than F.linear -- at [8,128] tokens an eager calibrating forward was 12.2 instead of 7.8 ms
        if x.is_cuda and x.numel() // self.in_features * self.in_features * self.out_features < options.INT8_CALIBRATION_MIN_MACS:
            return None
        pre = self._int8_forward(x, with_output_quantizer=False)     # bias + activation function applied, no quantizer
        if pre is None:
            return None
        self._save('', pre)
        out = amgr.quantize(pre)
        self._save('_Q', out)
        return out

    # ---- integer path ---------------------------------------------------------------------------
    def _int8_weights(self):
        """(int8 indices [N, K], int32 row sums [N], signed grid?) of the fake-quantized weight.  The indices are
        cached per (weight version, range state); the `signed` flag -- the one host read of this path -- per range
        state only, so that a training step (weights change every iteration, ranges fixed or learnable) re-quantizes
        its weights without a host synchronisation and stays hipGraph-capturable.  While a TRAINING step is being
        captured the weights are always re-quantized: the recorded launches must not depend on a cache hit."""
        wq = self.weight_quantizer.quantizer
        rkey = wq.range_state_key()
        if self._int8_signed is None or self._int8_signed[0] != rkey:
            self._int8_signed = (rkey, bool(wq.signed))          # host sync, once per range state
        key = (self.weight.data_ptr(), self.weight._version, rkey)
        recording = (torch.is_grad_enabled() and self.weight.requires_grad and torch.cuda.is_available()
                     and torch.cuda.is_current_stream_capturing())
        if self._int8_cache is None or self._int8_cache[0] != key or recording:
            be = _hip.backend()
            n_par = wq._delta.numel()
            w_idx = be.quantize_to_int8(self.weight.detach(), wq._delta, None, wq._signed, wq.n_bits, True,
                                        False, wq.eps, n_par, self.in_features if n_par > 1 else 1,
                                        minus_128=False)
            self._int8_cache = (key, w_idx, be.rowsum_i8(w_idx))
        return self._int8_cache[1], self._int8_cache[2], self._int8_signed[1]

    def _int8_weight_side_ok(self):
        """The weight half of the integer path's preconditions: fixed symmetric <= 8-bit linear-domain weight ranges
        (per-tensor or per-output-channel), weights quantized, nothing recording this layer's activations."""
        from quantization.quantization_manager import Qstates
        wmgr = self.weight_quantizer
        if _hooked(wmgr, getattr(wmgr, 'quantizer', None), self._modules.get('activation_quantizer'),
                   getattr(self._modules.get('activation_quantizer'), 'quantizer', None), self.activation_function):
            return False                 # somebody observes a stage the fused launch would skip: layered route
        if not (self._quant_w and self.activation_save_target is None and isinstance(wmgr, QuantizationManager)):
            return False
        if wmgr.state != Qstates.fix_ranges:
            # calibrating forward (options.INT8_CALIBRATION): the weight range is whatever the layered route's first step
            # -- get_params(): estimate, fake-quantize, cache in eval mode -- leaves behind; run exactly that step
            if not (options.INT8_CALIBRATION and not self.training and wmgr._estimating() and not torch.is_grad_enabled()):
                return False
            self.get_params()
        return bool(wmgr.quantizer.is_initialized
                    and wmgr.quantizer.symmetric and wmgr.quantizer.n_bits <= 8
                    and wmgr.quantizer.scale_domain == 'linear'
                    and wmgr.quantizer._delta.numel() in (1, self.out_features)
                    and not wmgr.quantizer._delta.requires_grad)

    def _int8_plan(self, x, with_output_quantizer=True):
        """Arguments of the integer evaluation of this layer for input `x`, or None when the configuration does not
        allow it (no fixed per-tensor asymmetric <= 8-bit input quantizer known for x, unsupported weight / output
        quantizer, shapes the MFMA kernel does not tile, ...)."""
        src = provenance.quantizer_of(x)                 # the quantizer that produced x (fixed range)
        if not _hip.on_device(x) or x.dtype != torch.float32:
            return None
        return self._int8_plan_from(src, x.numel() // self.in_features, with_output_quantizer)

    def _int8_plan_from(self, src, M, with_output_quantizer=True):
        """_int8_plan for an input that is known only by the quantizer `src` that produced it and its row count `M`
        (index-only producers: the fp32 tensor never exists)."""
        act_code = _ACT_CODES.get(type(self.activation_function))
        if (src is None or act_code is None or not self._int8_weight_side_ok()
                or src.symmetric or src.n_bits > 8 or src._delta is None or src._delta.numel() != 1
                or src.scale_domain != 'linear' or src._delta.requires_grad):
            return None
        if self.in_features % 64 or self.out_features % 32 or M % 32 or self.in_features > 16384:
            return None
        q_out = None
        amgr = self.activation_quantizer
        if with_output_quantizer and self._quant_a and not isinstance(amgr, FP32Acts):
            if not _fixed_per_tensor_manager(amgr) or amgr.quantizer._delta.requires_grad:
                return None
            oq = amgr.quantizer
            q_out = (oq._delta, oq._zero_float, getattr(oq, '_signed', None), oq.n_bits, oq.symmetric,
                     oq.scale_domain == 'log', oq.eps)
        return src, act_code, q_out

    def _int8_operands(self, x, plan, x_idx=None):
        """Kernel operands of the integer evaluation: (x_idx, w_idx, rowsum, bias, x_q, w_delta, w_eps), or None when
        the weight grid is unsigned (indices do not fit int8: the layered path runs, counted in INT8_STATS).
        x_idx given (index-only producer): `x` is not touched."""
        src = plan[0]
        be = _hip.backend()
        w_idx, rowsum, w_signed = self._int8_weights()
        if not w_signed:
            INT8_STATS['unsigned_weight_fallbacks'] += 1
            return None
        if x_idx is None:
            x_idx = provenance.indices_of(x)       # emitted by the producing quantizer in the same launch
        if x_idx is None or (x is not None and x_idx.shape != x.shape):
            x_idx = be.quantize_to_int8(x.detach(), src._delta, src._zero_float, None, src.n_bits, False, False,
                                        src.eps, 1, 1, minus_128=True)
        wq = self.weight_quantizer.quantizer
        bias = None if self.bias is None else self.bias.detach()
        return (x_idx, w_idx, rowsum, bias, (src._delta, src._zero_float, src.n_bits, src.eps), wq._delta.reshape(-1),
                wq.eps)

    def _int8_act_stair(self, act_code, q_out, rows):
        """(table, n_bins) of GELU + this layer's output quantizer for the integer epilogue of a call with `rows` input
        rows, cached per range state of that quantizer and bin count (built by one launch, no host read), or None (no
        GELU, no output quantizer, > 8 bits, switched off).  Unlike the int8 weights it is NOT rebuilt while a training
        step is being recorded: the integer path requires FIXED output ranges (`_int8_plan_from`), so the table a recorded
        launch reads stays valid for every replay, and `GraphedForward` / `GraphedTrainStep` keep the tensors of all
        derived caches they recorded alive (`quantization.graphs.derived_cache_tensors`)."""
        be = _hip.backend()
        if (act_code != _hip.ACT_GELU or q_out is None or q_out[3] > 8 or not options.INT8_ACT_STAIR
                or not hasattr(be, 'act_stair')):
            return None
        n_bins = be.stair_bins_for(rows, self.out_features) if hasattr(be, 'stair_bins_for') else None
        oq = self.activation_quantizer.quantizer
        # the table bakes in delta, zero_float, the grid ends and eps: every one of them is part of the key (an in-place
        # `zero_float.fill_()` moves neither `_range_gen` nor `_delta._version`)
        zf, sg = oq._buffers.get('_zero_float'), oq._buffers.get('_signed')
        key = (oq.range_state_key(), None if zf is None else zf._version, None if sg is None else sg._version, oq.eps,
               oq.symmetric)
        if self._int8_stair is None:
            self._int8_stair = {}
        cached = self._int8_stair.get(n_bins)           # one table per bin count (two call shapes may alternate)
        if (cached is None or cached[0] != key
                or getattr(cached[1][0], 'device', q_out[0].device) != q_out[0].device):
            cached = self._int8_stair[n_bins] = (key, be.act_stair(act_code, q_out, n_bins))
        return cached[1]

    def _int8_compute(self, x, plan, x_idx=None, index_only=False):
        """The fused integer Linear itself (no autograd): y [, its int8 indices] or None (unsigned weight grid).
        index_only: only the int8 indices of the output are produced and returned (the consumer is another integer
        Linear; needs an asymmetric <= 8-bit output quantizer in the plan)."""
        _, act_code, q_out = plan
        ops = self._int8_operands(x, plan, x_idx)
        if ops is None:
            return None
        amgr = self.activation_quantizer
        want_idx = q_out is not None and not amgr.quantizer.symmetric and amgr.quantizer.n_bits <= 8
        INT8_STATS['kernel_calls'] += 1
        stair = self._int8_act_stair(act_code, q_out, ops[0].numel() // self.in_features)
        if index_only:
            assert want_idx, 'index-only output needs an asymmetric <= 8-bit output quantizer'
            return _hip.backend().linear_i8(*ops[:5], ops[5], ops[6], act_code, q_out, torch.float32, want_idx=True,
                                            want_y=False, stair=stair)[1]
        out = _hip.backend().linear_i8(*ops[:5], ops[5], ops[6], act_code, q_out, torch.float32, want_idx=want_idx, stair=stair)
        y = out[0] if want_idx else out
        if q_out is not None:
            provenance.tag(y, amgr.quantizer, out[1] if want_idx else None)   # the next integer Linear consumes these
        return y

    def _int8_forward(self, x, with_output_quantizer=True):
        """Integer-GEMM evaluation of this layer, or None when the configuration does not allow it.
        with_output_quantizer=False returns the pre-quantizer output (for fused layer tails).

        Inference: the fused kernel alone.  Training / autograd (QAT with fixed ranges): the same integer forward on
        the matrix cores, wrapped in `_Int8LinearSTE` whose backward is the straight-through estimator of the layered
        modules (reference hijacker.py:66-116, quantizers.py:12-33)."""
        plan = self._int8_plan(x, with_output_quantizer)
        if plan is None:
            return None
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad or
                                                   (self.bias is not None and self.bias.requires_grad))
        if not needs_grad:
            return self._int8_compute(x, plan)
        if not with_output_quantizer:
            return None                      # fused tails are inference-only
        y = _Int8LinearSTE.apply(x, self.weight, self.bias, self, plan)
        if y is not None and plan[2] is not None:
            provenance.tag(y, self.activation_quantizer.quantizer, provenance.indices_of(y))
        return y


class _Int8LinearSTE(torch.autograd.Function):
    """y = Q_out(act(F.linear(x, Q_w(W), b))) evaluated exactly on the integer grids by the MFMA kernel; gradients of
    the layered modules.  The backward re-runs the layered forward of this ONE layer under autograd (fp32 GEMM, STE
    fake-quant kernels) and differentiates it: by construction the gradients w.r.t. x, W and b are those of the
    reference's module chain, at the cost of one recomputed GEMM (activation-checkpoint style); the pre-activation
    tensor never has to be written by the forward."""

    @staticmethod
    def forward(ctx, x, weight, bias, layer, plan):
        INT8_STATS['autograd_calls'] += 1
        y = layer._int8_compute(x, plan)
        if y is None:
            raise _hip.TQError('integer Linear: unsigned weight grid under autograd (disable options.INT8_LINEAR)')
        ctx.layer = layer
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        (x,) = ctx.saved_tensors
        layer = ctx.layer
        with torch.enable_grad():
            xr = x.detach().requires_grad_(True)
            y = QuantizationHijacker.forward(layer, xr)                 # layered path, straight-through estimators
            params = [p for p in (layer.weight, layer.bias) if p is not None and p.requires_grad]
            grads = torch.autograd.grad(y, [xr] + params, grad_y, allow_unused=True)
        gx = grads[0] if ctx.needs_input_grad[0] else None
        it = iter(grads[1:])
        gw = next(it) if layer.weight.requires_grad else None
        gb = next(it) if (layer.bias is not None and layer.bias.requires_grad) else None
        return gx, gw, gb, None, None


class QuantLayerNorm(QuantizationHijacker, nn.LayerNorm):
    def __init__(self, *args, activation=None, **kwargs):
        super().__init__(*args, activation=activation, **kwargs)

    def run_forward(self, x, weight, bias, offsets=None):
        return F.layer_norm(input=x.contiguous(), normalized_shape=self.normalized_shape,
                            weight=weight.contiguous(), bias=bias.contiguous(), eps=self.eps)


class QuantEmbedding(QuantizationHijacker, nn.Embedding):
    def __init__(self, *args, activation=None, **kwargs):
        super().__init__(*args, activation=activation, **kwargs)
        # a lookup in an already-quantized table needs no output quantizer
        self.activation_quantizer = FP32Acts()

    def run_forward(self, x, weight, bias, offsets=None):
        return F.embedding(input=x.contiguous(), weight=weight.contiguous(),
                           padding_idx=self.padding_idx, max_norm=self.max_norm,
                           norm_type=self.norm_type, scale_grad_by_freq=self.scale_grad_by_freq,
                           sparse=self.sparse)


class QuantNoNorm(QuantizationHijacker):
    """MobileBERT's element-wise affine "LayerNorm" with quantized parameters and output
    (counterpart of the reference's models/quantized_mobilebert.py:58-72).

    One weight quantizer is applied to the weight and then to the bias; while it is estimating,
    the bias call therefore overwrites the range found for the weight (upstream quirk, kept).
    With fixed ranges the whole layer -- x * Q(w) + Q(b) followed by the output quantizer -- is one
    fused launch (``tq_affine_fake_quant_fwd``) instead of ~8 element-wise kernels."""

    def __init__(self, org_model, *args, activation=None, **kwargs):
        super().__init__(*args, activation=activation, **kwargs)
        self.weight = org_model.weight
        self.bias = org_model.bias

    def _fusable(self, x):
        from quantization.quantization_manager import Qstates
        mgr = self.activation_quantizer
        return (self._quant_a and self.activation_function is None
                and self.activation_save_target is None
                and isinstance(mgr, QuantizationManager) and mgr.state == Qstates.fix_ranges
                and mgr.quantizer.is_initialized and mgr.quantizer._delta.numel() == 1
                and not torch.is_grad_enabled() and x.dim() >= 1
                and x.shape[-1] == self.weight.numel() and x.shape[-1] % 8 == 0)

    def quantized_params(self):
        """(Q(weight), Q(bias)) through the ONE weight quantizer, weight first (upstream order).  With fixed ranges in
        inference the pair is constant: cached per (parameter versions, range state) -- the reference re-quantizes both
        vectors on every forward (2 launches per NoNorm, 7 NoNorms per MobileBERT layer)."""
        if not self._quant_w:
            return self.weight, self.bias
        from quantization.quantization_manager import Qstates
        mgr = self.weight_quantizer
        cacheable = (not self.training and not torch.is_grad_enabled() and isinstance(mgr, QuantizationManager)
                     and mgr.state == Qstates.fix_ranges and mgr.quantizer.is_initialized)
        key = None
        if cacheable:
            key = (self.weight.data_ptr(), self.weight._version, self.bias.data_ptr(), self.bias._version,
                   mgr.quantizer.range_state_key())
            hit = getattr(self, '_qparam_cache', None)
            if hit is not None and hit[0] == key:
                return hit[1], hit[2]
        weight = mgr(self.weight)
        bias = mgr(self.bias)
        if cacheable:
            self._qparam_cache = (key, weight.detach(), bias.detach())
        return weight, bias

    def forward(self, x, offsets=None):
        weight, bias = self.quantized_params()
        if self._fusable(x):
            q = self.activation_quantizer.quantizer
            # feeding integer Linears (MobileBERT's bottlenecks -> query / key): emit the int8 indices in the same launch
            want_idx = (options.int8_active() and not q.symmetric and q.n_bits <= 8 and q.scale_domain == 'linear'
                        and _hip.on_device(x) and x.dtype == torch.float32)
            out = _hip.backend().affine_fake_quant(
                x, weight, bias, q._delta, q._zero_float, getattr(q, '_signed', None), q.n_bits,
                q.symmetric, q.scale_domain == 'log', q.eps, **({'want_idx': True} if want_idx else {}))
            if want_idx:
                return provenance.tag(out[0], q, out[1])
            return out
        return self.quantize_activations(x * weight + bias)


class QuantizedActivationWrapper(QuantizedActivation):
    """Runs `layer` and quantizes its output; can share ("tie") the quantizer of the layer
    before it, in which case the range is not updated here (useful for average pooling)."""

    def __init__(self, layer, tie_activation_quantizers=False,
                 input_quantizer: QuantizationManager = None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.tie_activation_quantizers = tie_activation_quantizers
        if input_quantizer:
            assert isinstance(input_quantizer, QuantizationManager)
            self.activation_quantizer = input_quantizer
        self.layer = layer

    def quantize_activations_no_range_update(self, x):
        return self.activation_quantizer.quantizer(x) if self._quant_a else x

    def forward(self, x):
        x = self.layer(x)
        if self.tie_activation_quantizers:
            return self.quantize_activations_no_range_update(x)
        return self.quantize_activations(x)


module_map = {nn.Linear: QuantLinear, nn.LayerNorm: QuantLayerNorm, nn.Embedding: QuantEmbedding}

non_param_modules = (_AdaptiveAvgPoolNd, _AvgPoolNd)


def get_act(module, i):
    """First activation function after position i of a Sequential, and its index."""
    for j in range(i + 1, len(module)):
        if isinstance(module[j], tuple(activations_list)):
            return module[j], j
    return None, None


def get_linear_args(module):
    return dict(in_features=module.in_features, out_features=module.out_features,
                bias=module.bias is not None)


def get_layernorm_args(module):
    return dict(normalized_shape=module.normalized_shape, eps=module.eps)


def get_embedding_args(module):
    return {k: getattr(module, k) for k in (
        'num_embeddings', 'embedding_dim', 'padding_idx', 'max_norm', 'norm_type',
        'scale_grad_by_freq', 'sparse')}


def get_module_args(mod, act):
    if isinstance(mod, nn.Linear):
        kwargs = get_linear_args(mod)
    elif isinstance(mod, nn.LayerNorm):
        kwargs = get_layernorm_args(mod)
    elif isinstance(mod, nn.Embedding):
        kwargs = get_embedding_args(mod)
    else:
        raise ValueError
    kwargs['activation'] = act
    return kwargs


def quant_module(module, i, **quant_params):
    """Quantized copy of module[i], folding a following activation function into it."""
    act, _ = get_act(module, i)
    src = module[i]
    new_module = module_map[type(src)](**get_module_args(src, act), **quant_params)
    new_module.weight.data = src.weight.data.clone()
    if src.bias is not None:
        new_module.bias.data = src.bias.data.clone()
    return new_module, i + int(bool(act)) + 1


def quantize_sequence(model, specials=None, tie_activation_quantizers=False, **quant_params):
    specials = specials or dict()
    out = []
    i = 0
    while i < len(model):
        m = model[i]
        if isinstance(m, QuantizedModule):
            out.append(m)
        elif type(m) in module_map:
            new_module, i = quant_module(model, i, **quant_params)
            out.append(new_module)
            continue
        elif type(m) in specials:
            out.append(specials[type(m)](m, **quant_params))
        elif isinstance(m, non_param_modules):
            input_quantizer = None
            if out and isinstance(out[-1], QuantizedModule) and tie_activation_quantizers:
                input_quantizer = out[-1].activation_quantizer
                warnings.warn(f'Tying input quantizer {i}^th layer of type {type(out[-1])} to the '
                              f'quantized {type(m)} following it')
            out.append(QuantizedActivationWrapper(
                m, tie_activation_quantizers=tie_activation_quantizers,
                input_quantizer=input_quantizer, **quant_params))
        else:
            out.append(quantize_model(m, specials=specials, **quant_params))
        i += 1
    return out


def quantize_sequential(model, specials=None, tie_activation_quantizers=False, **quant_params):
    return nn.Sequential(*quantize_sequence(model, specials, tie_activation_quantizers,
                                            **quant_params))


def quantize_module_list(model, specials=None, tie_activation_quantizers=False, **quant_params):
    return nn.ModuleList(quantize_sequence(model, specials, tie_activation_quantizers,
                                           **quant_params))


def quantize_model(model, specials=None, tie_activation_quantizers=False, **quant_params):
    """Recursively replace Linear / LayerNorm / Embedding (and `specials`) by quantized versions."""
    specials = specials or dict()

    if isinstance(model, nn.Sequential):
        return quantize_sequential(model, specials, tie_activation_quantizers, **quant_params)
    if type(model) in specials:
        return specials[type(model)](model, **quant_params)
    if isinstance(model, non_param_modules):
        return QuantizedActivationWrapper(model, **quant_params)
    if type(model) in module_map:
        # exact type match on purpose: subclasses of these layers are treated as containers
        quant_model = module_map[type(model)](**get_module_args(model, None), **quant_params)
        quant_model.weight.data = model.weight.data
        if getattr(model, 'bias', None) is not None:
            quant_model.bias.data = model.bias.data
        return quant_model

    quant_model = copy.deepcopy(model)
    for name, child in quant_model._modules.items():
        new_child = quantize_model(child, specials=specials, **quant_params)
        if new_child is not None:
            setattr(quant_model, name, new_child)
    return quant_model
