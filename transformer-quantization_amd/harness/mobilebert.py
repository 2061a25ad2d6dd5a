"""Synthetic-data MobileBERT harness on the drop-in quantization package (BASELINE config 4: MobileBERT W4A4).

The reference's models/quantized_mobilebert.py is out of scope as code (it wraps transformers-4.1 container forwards
that no longer exist), but its quantizer sites define the workload.  This module places the same quantizers at the same
tensor edges, in the same call order, around a plain re-statement of the MobileBERT forward (24 layers, hidden 512,
intra-bottleneck 128, 4 heads of 32, 4 stacked feed-forward networks, NoNorm everywhere, trigram embedding input),
using only the public API of `quantization/`.  Weights come from a random-init HuggingFace
`MobileBertForSequenceClassification` (no checkpoints offline).  Parity fixture: tests/golden/mobilebert_w4a4.npz,
produced by the reference's own blocks (tests/golden/make_golden_mobilebert.py).

Quantizer sites per layer, in call order (reference models/quantized_mobilebert.py):
  bottleneck.input (dense, NoNorm) :421-432 | bottleneck.attention (dense, NoNorm) | query, key, value :176-178 |
  scores :228, probs :238, context :250 | self-output dense, residual sum, NoNorm :258-304 |
  3 x FFN (intermediate+ReLU, dense, residual sum, NoNorm) :434-462 | intermediate+ReLU :481 |
  output dense, residual sum, NoNorm :361-405 | output bottleneck dense, residual sum, NoNorm :321-358
= 32 activation quantizers and 23 weight quantizers per layer; 774 + 559 in the whole model.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from quantization import options
from quantization.autoquant_utils import QuantNoNorm, quantize_model
from quantization.base_quantized_classes import FP32Acts, QuantizedActivation
from quantization.base_quantized_model import QuantizedModel
from quantization.fused import hooked as _hooked
from quantization.range_estimators import OptMethod, RangeEstimators

# per-site switches of the reference (DEFAULT_QUANT_DICT, models/quantized_mobilebert.py:31-49)
DEFAULT_QUANT_DICT = {
    'sum_input_pos_embd': True, 'sum_token_type_embd': True,
    'attn_scores': True, 'attn_probs': True, 'attn_probs_n_bits_act': None, 'attn_probs_act_range_method': None,
    'attn_probs_act_range_options': None, 'attn_output': True,
    'res_self_output': True, 'res_output': True, 'res_output_bottleneck': True, 'res_ffn_output': True,
}


def _site(enabled, **qp):
    return QuantizedActivation(**qp) if enabled else FP32Acts()


def _split_qp(quant_params):
    """(kwargs for the quantized modules, resolved per-site dict)."""
    qp = dict(quant_params)
    sites = dict(DEFAULT_QUANT_DICT)
    sites.update(qp.pop('quant_dict', None) or {})
    return qp, sites


class QMobileEmbeddings(QuantizedModel):
    def __init__(self, hf, **quant_params):
        super().__init__()
        qp, sites = _split_qp(quant_params)
        self.trigram_input = hf.trigram_input
        self.embedding_size, self.hidden_size = hf.embedding_size, hf.hidden_size
        self.word_embeddings = quantize_model(hf.word_embeddings, **qp)
        self.position_embeddings = quantize_model(hf.position_embeddings, **qp)
        self.token_type_embeddings = quantize_model(hf.token_type_embeddings, **qp)
        self.embedding_transformation = quantize_model(hf.embedding_transformation, **qp)
        self.LayerNorm = QuantNoNorm(hf.LayerNorm, **qp)
        self.sum_input_pos_embd_act_quantizer = _site(sites['sum_input_pos_embd'], **qp)
        self.sum_token_type_embd_act_quantizer = _site(sites['sum_token_type_embd'], **qp)

    def forward(self, input_ids):
        B, T = input_ids.shape
        e = self.word_embeddings(input_ids)                                        # [B, T, 128]
        if self.trigram_input:                                                     # kernel-3 "convolution" as a concat
            e = torch.cat([F.pad(e[:, 1:], [0, 0, 0, 1, 0, 0], value=0.0), e,
                           F.pad(e[:, :-1], [0, 0, 1, 0, 0, 0], value=0.0)], dim=2)   # [B, T, 384]
        if self.trigram_input or self.embedding_size != self.hidden_size:
            e = self.embedding_transformation(e)                                   # [B, T, 512]
        from harness.bert import constant
        pos = constant('positions', 1, T, input_ids.device)
        tok = constant('token_type', input_ids.shape[0], T, input_ids.device)
        x = self.sum_input_pos_embd_act_quantizer(e + self.position_embeddings(pos))
        x = self.sum_token_type_embd_act_quantizer(x + self.token_type_embeddings(tok))
        return self.LayerNorm(x)


class QBottleneckLayer(QuantizedModel):
    """dense -> NoNorm (512 -> 128)."""

    def __init__(self, hf, **qp):
        super().__init__()
        self.dense = quantize_model(hf.dense, **qp)
        self.LayerNorm = QuantNoNorm(hf.LayerNorm, **qp)

    fuse = None    # set True: dense + NoNorm + both quantizers as one integer launch (quantization/fused.py)

    def forward(self, h):
        if options.fuse_on(self.fuse, self, self.LayerNorm):
            from quantization.fused import linear_nonorm_quant
            return linear_nonorm_quant(self.dense, self.LayerNorm, h)
        return self.LayerNorm(self.dense(h))


class QMobileSelfAttention(QuantizedModel):
    def __init__(self, hf, **quant_params):
        super().__init__()
        qp, sites = _split_qp(quant_params)
        self.heads, self.head_dim = hf.num_attention_heads, hf.attention_head_size
        self.query = quantize_model(hf.query, **qp)
        self.key = quantize_model(hf.key, **qp)
        self.value = quantize_model(hf.value, **qp)
        self.attn_scores_act_quantizer = _site(sites['attn_scores'], **qp)
        probs_qp = dict(qp)
        if sites['attn_probs_n_bits_act'] is not None:
            probs_qp['n_bits_act'] = sites['attn_probs_n_bits_act']
        if sites['attn_probs_act_range_method'] is not None:
            probs_qp['act_range_method'] = RangeEstimators[sites['attn_probs_act_range_method']]
        if sites['attn_probs_act_range_options'] is not None:
            opts = dict(sites['attn_probs_act_range_options'])
            if 'opt_method' in opts and not isinstance(opts['opt_method'], OptMethod):
                opts['opt_method'] = OptMethod[opts['opt_method']]
            probs_qp['act_range_options'] = opts
        self.attn_probs_act_quantizer = _site(sites['attn_probs'], **probs_qp)
        self.attn_output_act_quantizer = _site(sites['attn_output'], **qp)

    def _split(self, x):
        B, T, _ = x.shape
        return x.view(B, T, self.heads, self.head_dim).permute(0, 2, 1, 3)

    fuse = None    # set True to run the fixed-range attention core as one integer kernel (quantization/fused.py)

    def forward(self, q_in, k_in, v_in, mask, value_out=None):
        """value_out: self.value(v_in) when the layer computed it along with its input bottlenecks (same input tensor)"""
        fuse = options.fuse_on(self.fuse, self, self.attn_probs_act_quantizer)
        if fuse:
            # the three Linears as index-only grouped integer launches (query and key share the bottlenecked input: one
            # launch) feeding the integer core directly; None = not eligible
            from quantization.fused import quantized_self_attention
            ctx = quantized_self_attention((q_in, k_in, v_in), self.query, self.key, self.value, mask, self.heads,
                                           self.attn_scores_act_quantizer, self.attn_probs_act_quantizer,
                                           self.attn_output_act_quantizer, value_out=value_out)
            if ctx is not None:
                return ctx
        qo, ko, vo = self.query(q_in), self.key(k_in), (self.value(v_in) if value_out is None else value_out)
        if fuse:
            # Q K^T -> quantizer -> / sqrt(d) + mask -> softmax -> quantizer -> P V -> quantizer (per-tensor, so "per head
            # before the merge" and "after the merge" coincide) on the i8 matrix cores; None = layered modules
            from quantization.fused import quantized_attention
            ctx = quantized_attention(qo, ko, vo, mask, self.heads, self.attn_scores_act_quantizer,
                                      self.attn_probs_act_quantizer, self.attn_output_act_quantizer)
            if ctx is not None:
                return ctx
        q, k, v = self._split(qo), self._split(ko), self._split(vo)
        scores = self.attn_scores_act_quantizer(torch.matmul(q, k.transpose(-1, -2)))
        scores = scores / math.sqrt(self.head_dim)
        if mask is not None:
            scores = scores + mask
        probs = self.attn_probs_act_quantizer(torch.softmax(scores, dim=-1))
        ctx = self.attn_output_act_quantizer(torch.matmul(probs, v))              # quantized per head, before the merge
        ctx = ctx.permute(0, 2, 1, 3).contiguous()
        return ctx.view(ctx.shape[0], ctx.shape[1], self.heads * self.head_dim)


class QResidualNoNorm(QuantizedModel):
    """dense -> (+ residual) -> quantize -> NoNorm: MobileBertSelfOutput / FFNOutput / MobileBertOutput /
    OutputBottleneck all have this shape.  `fuse = True` runs the fixed-range tail as ONE kernel
    (tq_residual_nonorm_quant_fwd, quantization/fused.py)."""

    fuse = None

    def __init__(self, hf, site_on=True, **qp):
        super().__init__()
        self.dense = quantize_model(hf.dense, **qp)
        self.res_act_quantizer = _site(site_on, **qp)
        self.LayerNorm = QuantNoNorm(hf.LayerNorm, **qp)

    def forward(self, h, residual):
        if options.fuse_on(self.fuse, self, self.LayerNorm):
            from quantization.fused import residual_layernorm_quant
            return residual_layernorm_quant(self.dense, self.res_act_quantizer, self.LayerNorm, h, residual)
        return self.LayerNorm(self.res_act_quantizer(self.dense(h) + residual))


class QFFN(QuantizedModel):
    def __init__(self, hf, **quant_params):
        super().__init__()
        qp, sites = _split_qp(quant_params)
        self.intermediate = quantize_model(nn.Sequential(hf.intermediate.dense, nn.ReLU()), **qp)
        self.output = QResidualNoNorm(hf.output, sites['res_ffn_output'], **qp)

    fuse = None    # set True: intermediate + output + NoNorm tail as one integer launch (quantization/fused.py quantized_ffn)

    def forward(self, h):
        # (the merged launch calls neither self.output nor the Sequential: forward hooks on those containers keep the layered route)
        if options.fuse_on(self.fuse, self, self.output.LayerNorm) and not _hooked(self.output, self.intermediate):
            return _ffn(self.intermediate, self.output, h)
        return self.output(self.intermediate(h), h)


def _ffn(intermediate, out_block, h):
    from quantization.fused import quantized_ffn
    return quantized_ffn(intermediate[0], out_block.dense, out_block.res_act_quantizer, out_block.LayerNorm, h)


class QMobileLayer(QuantizedModel):
    def __init__(self, hf, **quant_params):
        super().__init__()
        qp, sites = _split_qp(quant_params)
        assert hf.use_bottleneck and hf.bottleneck.key_query_shared_bottleneck and not hf.bottleneck.use_bottleneck_attention, \
            'harness covers the default MobileBertConfig (shared key/query bottleneck)'
        self.bottleneck_input = QBottleneckLayer(hf.bottleneck.input, **qp)
        self.bottleneck_attention = QBottleneckLayer(hf.bottleneck.attention, **qp)
        self.attention_self = QMobileSelfAttention(hf.attention.self, **quant_params)
        self.attention_output = QResidualNoNorm(hf.attention.output, sites['res_self_output'], **qp)
        self.ffn = nn.ModuleList([QFFN(f, **quant_params) for f in hf.ffn])
        self.intermediate = quantize_model(nn.Sequential(hf.intermediate.dense, nn.ReLU()), **qp)
        self.output = QResidualNoNorm(hf.output, sites['res_output'], **qp)
        self.output_bottleneck = QResidualNoNorm(hf.output.bottleneck, sites['res_output_bottleneck'], **qp)

    fuse_ffn = None    # set True: the last feed-forward block (intermediate + output) as one integer launch, like QFFN.fuse
    fuse_chain = True  # all feed-forward blocks of the layer as ONE launch when each of them is fused (False: one launch per block)

    def forward(self, h, mask):
        pair = None
        # Merged launches bypass the __call__ of the containers they replace (the two QBottleneckLayers; every QFFN, its
        # QResidualNoNorm and the Sequential around its intermediate Linear): a forward hook on one of them keeps the
        # un-merged modules, whose own helpers check the leaves.
        if (options.fuse_on(self.bottleneck_input.fuse, self.bottleneck_input, self.bottleneck_input.LayerNorm)
                and not _hooked(self.bottleneck_input, self.bottleneck_attention)):
            from quantization.fused import linear_nonorm_quant_pair     # both bottlenecks read h: one integer launch
            with_value = options.fuse_on(self.attention_self.fuse, self.attention_self, self.attention_self.attn_probs_act_quantizer)
            pair = linear_nonorm_quant_pair(self.bottleneck_input.dense, self.bottleneck_input.LayerNorm,
                                            self.bottleneck_attention.dense, self.bottleneck_attention.LayerNorm, h,
                                            value=self.attention_self.value if with_value else None)
        value_out = None
        if pair is not None:
            layer_input, shared = pair[0], pair[1]
            value_out = pair[2] if len(pair) == 3 else None       # the value Linear rode along (same input)
        else:
            layer_input = self.bottleneck_input(h)                # [B, T, 128] residual of the attention block
            shared = self.bottleneck_attention(h)                 # query / key input
        a = self.attention_output(self.attention_self(shared, shared, h, mask, value_out=value_out), layer_input)
        o = None
        if (options.fuse_on(self.fuse_ffn, self, self.output.LayerNorm) and self.fuse_chain
                and all(options.fuse_on(f.fuse, f, f.output.LayerNorm) for f in self.ffn)
                and not _hooked(self.intermediate, self.output, *self.ffn, *(f.output for f in self.ffn),
                                *(f.intermediate for f in self.ffn))):
            from quantization.fused import quantized_ffn_chain       # the four feed-forward blocks as ONE integer launch
            o = quantized_ffn_chain([(f.intermediate[0], f.output.dense, f.output.res_act_quantizer, f.output.LayerNorm)
                                     for f in self.ffn] + [(self.intermediate[0], self.output.dense,
                                                            self.output.res_act_quantizer, self.output.LayerNorm)], a)
        if o is None:
            for f in self.ffn:
                a = f(a)
            if options.fuse_on(self.fuse_ffn, self, self.output.LayerNorm) and not _hooked(self.intermediate, self.output):
                o = _ffn(self.intermediate, self.output, a)
            else:
                o = self.output(self.intermediate(a), a)
        return self.output_bottleneck(o, h)                       # back to 512, residual = the layer's input


class QMobileBertForSequenceClassification(QuantizedModel):
    def __init__(self, hf, quant_setup='all', **quant_params):
        super().__init__()
        qp, _ = _split_qp(quant_params)
        mb = hf.mobilebert
        self.embeddings = QMobileEmbeddings(mb.embeddings, **quant_params)
        self.layers = nn.ModuleList([QMobileLayer(l, **quant_params) for l in mb.encoder.layer])
        self.do_activate = mb.pooler.do_activate
        if self.do_activate:
            self.pooler = quantize_model(nn.Sequential(mb.pooler.dense, nn.Tanh()), **qp)
        self.classifier = quantize_model(hf.classifier, **qp)
        if quant_setup == 'FP_logits':
            self.classifier.activation_quantizer = FP32Acts()
        elif quant_setup not in (None, 'all'):
            raise ValueError("Quantization setup '{}' not supported.".format(quant_setup))

    def forward(self, input_ids, attention_mask=None):
        if attention_mask is not None:
            mask = (1.0 - attention_mask[:, None, None, :].float()) * -10000.0
        else:
            from harness.bert import constant
            mask = constant('mask', input_ids.shape[0], input_ids.shape[1], input_ids.device)
        h = self.embeddings(input_ids)
        for layer in self.layers:
            h = layer(h, mask)
        pooled = h[:, 0]
        if self.do_activate:
            pooled = self.pooler(pooled)
        return self.classifier(pooled)


def build_mobilebert(seed=1000, num_labels=2, num_layers=None, **qp):
    """HF MobileBERT architecture with parameters from the build-independent numpy stream (harness/weights.py: NoNorm
    gets non-trivial affine parameters, see there), wrapped with quantizers."""
    from transformers import MobileBertConfig, MobileBertForSequenceClassification
    from harness.weights import fill_from_numpy_stream
    torch.manual_seed(seed)
    cfg = MobileBertConfig(num_labels=num_labels)
    if num_layers is not None:
        cfg.num_hidden_layers = num_layers
    hf = fill_from_numpy_stream(MobileBertForSequenceClassification(cfg).eval(), seed)
    return QMobileBertForSequenceClassification(hf, **qp), hf
