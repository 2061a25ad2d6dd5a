"""Synthetic-data BERT harness built on the drop-in quantization package (used by the parity tests, the
benchmarks and the validate_quantized CLI).

The reference's models/quantized_bert.py is out of scope as code (it depends on transformers-4.1
container internals), but its tensor edges define the workload: 161 activation quantizers and 102
weight quantizers for BERT-base.  This module places quantizers at the same
edges, in the same order, around a plain re-statement of the BERT forward, using only the public API
of `quantization/` -- it is the consumer the parity test (tests/test_bert_e2e.py) and the whole-model
calibration benchmark drive.  Weights come from a HuggingFace `BertForSequenceClassification`
(random-init; no checkpoints are available offline).

Quantizer sites per encoder layer, in call order (reference models/quantized_bert.py:135-146, 154,
198, 213, 239-245, 265-277, 291): q, k, v | scores | probs | context | self-output dense | residual
sum | LayerNorm | intermediate(+GELU) | output dense | residual sum | LayerNorm.
"""
import math

import torch
from torch import nn

from quantization import options
from quantization.autoquant_utils import quantize_model
from quantization.base_quantized_classes import QuantizedActivation
from quantization.base_quantized_model import QuantizedModel
from quantization.fused import hooked as _hooked


_CONSTANTS = {}


def constant(kind, B, T, device):
    """Read-only helper tensors of a forward that depend on the batch shape only -- the all-visible additive attention
    mask, the all-zero token-type ids, the position ids 0 .. T-1 -- built once per (shape, device) instead of by a fill /
    arange launch in every forward (three ~5 us launches of a 0.84 ms hipGraph forward).  Never modified in place.
    Tensors created under torch.inference_mode() are inference tensors and cannot be saved for a later backward
    (F.embedding(position ids) of a training forward with the same shape would raise): they get cache entries of their own."""
    key = (kind, B, T, str(device), torch.is_inference_mode_enabled())
    t = _CONSTANTS.get(key)
    if t is None:
        if len(_CONSTANTS) > 64:
            _CONSTANTS.clear()
        if kind == 'mask':
            t = torch.zeros(B, 1, 1, T, device=device)
        elif kind == 'token_type':
            t = torch.zeros(B, T, dtype=torch.long, device=device)
        else:
            t = torch.arange(T, device=device).unsqueeze(0)
        _CONSTANTS[key] = t
    return t


class QEmbeddings(QuantizedModel):
    def __init__(self, hf, **qp):
        super().__init__()
        self.word_embeddings = quantize_model(hf.word_embeddings, **qp)
        self.position_embeddings = quantize_model(hf.position_embeddings, **qp)
        self.token_type_embeddings = quantize_model(hf.token_type_embeddings, **qp)
        self.sum_input_token_type_embd_act_quantizer = QuantizedActivation(**qp)
        self.sum_pos_embd_act_quantizer = QuantizedActivation(**qp)
        self.LayerNorm = quantize_model(hf.LayerNorm, **qp)

    def position_ids(self, input_ids):
        """BERT: 0 .. T-1 whatever the tokens (RoBERTa overrides this, harness/roberta.py)."""
        return constant('positions', 1, input_ids.shape[1], input_ids.device)

    fuse = None    # set True: the whole block as one kernel with fixed ranges (quantization/fused.py embeddings_layernorm_quant)

    def forward(self, input_ids):
        pos = self.position_ids(input_ids)
        tok = constant('token_type', input_ids.shape[0], input_ids.shape[1], input_ids.device)
        if options.fuse_on(self.fuse, self, self.sum_pos_embd_act_quantizer):
            from quantization.fused import embeddings_layernorm_quant
            y = embeddings_layernorm_quant(self.word_embeddings, self.token_type_embeddings, self.position_embeddings,
                                           self.sum_input_token_type_embd_act_quantizer, self.sum_pos_embd_act_quantizer,
                                           self.LayerNorm, input_ids, tok, pos)
            if y is not None:
                return y
        x = self.word_embeddings(input_ids) + self.token_type_embeddings(tok)
        x = self.sum_input_token_type_embd_act_quantizer(x)
        x = x + self.position_embeddings(pos)
        x = self.sum_pos_embd_act_quantizer(x)
        return self.LayerNorm(x)


class QSelfAttention(QuantizedModel):
    def __init__(self, hf, **qp):
        super().__init__()
        self.heads, self.head_dim = hf.num_attention_heads, hf.attention_head_size
        self.query = quantize_model(hf.query, **qp)
        self.key = quantize_model(hf.key, **qp)
        self.value = quantize_model(hf.value, **qp)
        self.attn_scores_act_quantizer = QuantizedActivation(**qp)
        self.attn_probs_act_quantizer = QuantizedActivation(**qp)
        self.context_act_quantizer = QuantizedActivation(**qp)

    def _split(self, x):
        B, T, _ = x.shape
        return x.view(B, T, self.heads, self.head_dim).permute(0, 2, 1, 3)

    fuse = None    # set True: scores-quant -> scale -> mask -> softmax -> probs-quant as one kernel

    def forward(self, h, mask):
        fused = options.fuse_on(self.fuse, self, self.attn_probs_act_quantizer)
        if fused:
            from quantization.fused import quantized_self_attention
            ctx = quantized_self_attention(h, self.query, self.key, self.value, mask, self.heads,
                                           self.attn_scores_act_quantizer, self.attn_probs_act_quantizer,
                                           self.context_act_quantizer)
            if ctx is not None:                  # stacked QKV projection + attention core: 2 integer kernels
                return ctx
        qo, ko, vo = self.query(h), self.key(h), self.value(h)
        if fused:
            from quantization.fused import quantized_attention
            ctx = quantized_attention(qo, ko, vo, mask, self.heads, self.attn_scores_act_quantizer,
                                      self.attn_probs_act_quantizer, self.context_act_quantizer)
            if ctx is not None:                  # whole attention core as one integer kernel
                return ctx
        q, k, v = self._split(qo), self._split(ko), self._split(vo)
        raw = torch.matmul(q, k.transpose(-1, -2))
        if fused:
            from quantization.fused import scores_softmax_quant
            probs = scores_softmax_quant(self.attn_scores_act_quantizer, self.attn_probs_act_quantizer, raw,
                                         mask, math.sqrt(self.head_dim))
        else:
            scores = self.attn_scores_act_quantizer(raw)
            scores = scores / math.sqrt(self.head_dim)
            if mask is not None:
                scores = scores + mask
            probs = self.attn_probs_act_quantizer(torch.softmax(scores, dim=-1))
        ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).contiguous()
        # explicit width: an empty batch (a rank without calibration samples) has no elements to infer it from
        return self.context_act_quantizer(ctx.view(ctx.shape[0], ctx.shape[1], self.heads * self.head_dim))


class QResidualBlock(QuantizedModel):
    """dense -> (+ residual) -> quantize -> LayerNorm (BertSelfOutput / BertOutput)."""

    def __init__(self, hf, **qp):
        super().__init__()
        self.dense = quantize_model(hf.dense, **qp)
        self.res_act_quantizer = QuantizedActivation(**qp)
        self.LayerNorm = quantize_model(hf.LayerNorm, **qp)

    fuse = None    # set True to run the fixed-range tail as one kernel (quantization/fused.py)

    def forward(self, h, residual):
        if options.fuse_on(self.fuse, self, self.res_act_quantizer):
            from quantization.fused import residual_layernorm_quant
            return residual_layernorm_quant(self.dense, self.res_act_quantizer, self.LayerNorm, h, residual)
        return self.LayerNorm(self.res_act_quantizer(self.dense(h) + residual))


class QLayer(QuantizedModel):
    def __init__(self, hf, **qp):
        super().__init__()
        self.attention_self = QSelfAttention(hf.attention.self, **qp)
        self.attention_output = QResidualBlock(hf.attention.output, **qp)
        self.intermediate = quantize_model(nn.Sequential(hf.intermediate.dense, nn.GELU()), **qp)
        self.output = QResidualBlock(hf.output, **qp)

    fuse_ffn = None    # set True (with options.INT8_LINEAR): intermediate runs index-only, its [B, T, 3072] fp32 output is
                       # never stored (quantization/fused.py quantized_bert_ffn)

    def forward(self, h, mask):
        a = self.attention_output(self.attention_self(h, mask), h)
        # (the merged launch calls neither self.output nor the Sequential: forward hooks on those containers keep the layered route)
        if options.fuse_on(self.fuse_ffn, self, self.output.res_act_quantizer) and not _hooked(self.output, self.intermediate):
            from quantization.fused import quantized_bert_ffn
            out = self.output
            return quantized_bert_ffn(self.intermediate[0], out.dense, out.res_act_quantizer, out.LayerNorm, a, a)
        return self.output(self.intermediate(a), a)


class QBertForSequenceClassification(QuantizedModel):
    """quant_setup (reference models/quantized_bert.py:526-555): 'all' quantizes the logits like every
    other activation, 'FP_logits' leaves them in fp32, 'MSE_logits' estimates their range with the
    golden-section MSE estimator."""

    def __init__(self, hf, quant_setup='all', **qp):
        super().__init__()
        self.embeddings = QEmbeddings(hf.bert.embeddings, **qp)
        self.layers = nn.ModuleList([QLayer(l, **qp) for l in hf.bert.encoder.layer])
        self.pooler = quantize_model(nn.Sequential(hf.bert.pooler.dense, nn.Tanh()), **qp)
        head_qp = dict(qp)
        if quant_setup == 'MSE_logits':
            from quantization.range_estimators import OptMethod, RangeEstimators
            head_qp['act_range_method'] = RangeEstimators.MSE
            head_qp['act_range_options'] = dict(opt_method=OptMethod.golden_section)
        elif quant_setup not in ('all', 'FP_logits'):
            raise ValueError("Quantization setup '{}' not supported.".format(quant_setup))
        self.classifier = quantize_model(hf.classifier, **head_qp)
        if quant_setup == 'FP_logits':
            from quantization.base_quantized_classes import FP32Acts
            self.classifier.activation_quantizer = FP32Acts()

    def forward(self, input_ids, attention_mask=None):
        mask = None
        if attention_mask is not None:
            mask = (1.0 - attention_mask[:, None, None, :].float()) * -10000.0
        else:
            mask = constant('mask', input_ids.shape[0], input_ids.shape[1], input_ids.device)
        h = self.embeddings(input_ids)
        for layer in self.layers:
            h = layer(h, mask)
        return self.classifier(self.pooler(h[:, 0]))


def build_bert_base(seed=1000, num_labels=2, num_layers=None, **qp):
    """HF BERT-base architecture with parameters from the build-independent numpy stream (harness/weights.py),
    wrapped with quantizers."""
    from transformers import BertConfig, BertForSequenceClassification
    from harness.weights import fill_from_numpy_stream
    torch.manual_seed(seed)
    cfg = BertConfig(num_labels=num_labels)
    if num_layers is not None:
        cfg.num_hidden_layers = num_layers
    hf = fill_from_numpy_stream(BertForSequenceClassification(cfg).eval(), seed)
    return QBertForSequenceClassification(hf, **qp), hf


def quantizer_census(model):
    from quantization.quantization_manager import QuantizationManager
    act = [(n, m) for n, m in model.named_modules()
           if isinstance(m, QuantizationManager) and n.endswith('activation_quantizer')]
    wts = [(n, m) for n, m in model.named_modules()
           if isinstance(m, QuantizationManager) and n.endswith('weight_quantizer')]
    return act, wts


# ---- per-embedding(-group) / per-token wiring and mixed-precision overrides -------------------------
# The reference hard-codes these walks over `model.bert...` in main.py:358-439 (axis / groups) and
# main.py:442-500 (the `--quant-dict` letter codes); here they are tables over this harness's modules.

def _embd_sites(model, with_pooler):
    E = model.embeddings
    sites = [E.sum_input_token_type_embd_act_quantizer, E.sum_pos_embd_act_quantizer, E.LayerNorm]
    for L in model.layers:
        A, S, O = L.attention_self, L.attention_output, L.output
        sites += [A.query, A.key, A.value, A.context_act_quantizer,
                  S.dense, S.res_act_quantizer, S.LayerNorm, O.dense, O.res_act_quantizer, O.LayerNorm]
    return sites, ([model.pooler[0]] if with_pooler else [])    # pooler = Sequential(QuantLinear+tanh)


def apply_activation_granularity(model, per_token=False, per_embd=False, per_groups=None, permute=False):
    """`--per-token / --per-embd / --per-groups N / --per-groups-permute`: [B, T, d] tensors are quantized
    along axis 2 (per embedding dimension, optionally folded into N groups) or axis 1 (per token); the
    [B, d] pooler output along axis 1 for per-embedding only.  Attention scores / probabilities and the
    [B, T, 3072] intermediate keep per-tensor ranges, as upstream."""
    from utils.per_embd_quant_utils import set_act_quant_axis_and_groups
    if not (per_token or per_embd or per_groups):
        return 0
    axis = 2 if (per_embd or per_groups) else 1
    sites, pooled = _embd_sites(model, with_pooler=per_embd)
    for m in sites:
        set_act_quant_axis_and_groups(m, axis=axis, n_groups=per_groups, permute=permute)
    for m in pooled:
        set_act_quant_axis_and_groups(m, axis=1, n_groups=per_groups, permute=permute)
    return len(sites) + len(pooled)


def apply_quant_dict(model, quant_dict):
    """`--quant-dict "{'y': 'ng6', 'h': 16, 'Et': 4, ...}"`: per-site overrides keyed by the reference's
    letter codes, optionally suffixed with a layer index (main.py:442-500)."""
    from utils.per_embd_quant_utils import hijack_act_quant, hijack_act_quant_modules, hijack_weight_quant
    if not quant_dict:
        return
    E = model.embeddings
    for m in (E.sum_input_token_type_embd_act_quantizer, E.sum_pos_embd_act_quantizer):
        hijack_act_quant(quant_dict, 'e', m)
    hijack_weight_quant(quant_dict, 'Et', E.word_embeddings)
    for i, L in enumerate(model.layers):
        A, S, O = L.attention_self, L.attention_output, L.output
        table = (('s', A.attn_scores_act_quantizer), ('p', A.attn_probs_act_quantizer),
                 ('c', A.context_act_quantizer), ('g', S.dense), ('u', S.res_act_quantizer),
                 ('x', S.LayerNorm), ('h', O.dense), ('y', O.res_act_quantizer), ('z', O.LayerNorm))
        for code, m in table:
            hijack_act_quant(quant_dict, f'{code}{i}', m)
            hijack_act_quant(quant_dict, code, m)
        hijack_act_quant_modules(quant_dict, f'L{i}', L)
        hijack_act_quant_modules(quant_dict, 'L', L)
    hijack_act_quant(quant_dict, 'P', model.pooler[0])
    hijack_act_quant(quant_dict, 'C', model.classifier)
    hijack_weight_quant(quant_dict, 'wP', model.pooler[0])
    hijack_weight_quant(quant_dict, 'wC', model.classifier)


def estimate_permutation_ranges(model, batches, shared_h=False):
    """Phase 1 of `--per-groups-permute` (main.py:512-560): an FP32 pass that only collects per-dimension
    ranges, from which each estimator derives its range-sorted group assignment; with `shared_h` all
    estimators of a layer reuse the ranges seen at the FFN-output dense (one permutation per layer)."""
    from quantization.range_estimators import RangeEstimatorBase
    from utils.utils import pass_data_for_range_estimation
    model.full_precision()
    pass_data_for_range_estimation(loader=batches, model=model, act_quant=True, weight_quant=False,
                                   max_num_batches=10, cross_entropy_layer=None)
    for m in model.modules():
        if isinstance(m, RangeEstimatorBase):
            m.per_group_range_estimation = False
    if shared_h:
        for L in model.layers:
            ests = {n: m for n, m in L.named_modules() if isinstance(m, RangeEstimatorBase) and m.ranges is not None}
            src = [m.ranges for n, m in ests.items() if 'output.dense' in n]
            assert src, 'no FFN-output range estimator with collected ranges in this layer'
            for m in ests.values():
                m.ranges = src[-1].clone()
