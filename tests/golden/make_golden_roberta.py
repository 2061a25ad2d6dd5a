#!/usr/bin/env python3
"""RoBERTa fixture: a random-init 2-layer RoBERTa-base, W8A8 per-tensor (symmetric 8-bit weights / current min-max,
asymmetric 8-bit activations / running min-max), one calibration batch, then a fixed-range forward -- produced by the
REFERENCE's own models/quantized_roberta.py blocks, imported here (build container only).

What RoBERTa adds to the BERT fixture (make_golden_bert.py): position ids derived from the input ids (padding tokens
keep position `padding_idx`, the others count from `padding_idx + 1`; models/quantized_roberta.py:26-41, 63-104), an
attention mask with real padding, no pooler, and the classification head (dense -> tanh -> out_proj on the first
token) quantized by the generic recursive rewriter (`quantize_model(org_model.classifier)`, :158): its tanh is a
functional call, so the dense's output quantizer sees the pre-tanh values and out_proj consumes un-quantized tanh output.

As for BERT the HF 4.1 container forwards no longer match today's transformers, so the script drives the reference's
quantized blocks in the order the containers did.  Same shims as make_golden_bert.py; no reference file is modified.

    python tests/golden/make_golden_roberta.py        -> tests/golden/roberta_2l_w8a8.npz
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = '/root/reference'
sys.path.insert(0, REF)
_u = types.ModuleType('utils')
_u.__path__ = [os.path.join(REF, 'utils')]
for _name in ('_tb_advance_global_step', '_tb_advance_token_counters', '_tb_hist'):
    setattr(_u, _name, lambda *a, **k: None)
sys.modules['utils'] = _u

import numpy as np  # noqa: E402
import torch  # noqa: E402
import transformers  # noqa: E402
import transformers.modeling_utils as _mu  # noqa: E402
from transformers.pytorch_utils import apply_chunking_to_forward  # noqa: E402
_mu.apply_chunking_to_forward = apply_chunking_to_forward
from transformers import RobertaConfig, RobertaForSequenceClassification  # noqa: E402

from utils.utils import DotDict  # noqa: E402
_u.DotDict = DotDict
from quantization.quantizers import QMethods  # noqa: E402
from quantization.range_estimators import RangeEstimators  # noqa: E402
from quantization.quantization_manager import QuantizationManager  # noqa: E402
from quantization.autoquant_utils import quantize_model  # noqa: E402
from models.quantized_roberta import QuantizedRobertaEmbeddings, QuantizedRobertaLayer  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SEED = 1000

# build-independent parameters: the SAME function the harness models use (numpy + torch only, loaded by path so that the
# repo's `quantization` package is never imported next to the reference's)
import importlib.util as _ilu  # noqa: E402
_spec = _ilu.spec_from_file_location('tq_harness_weights', os.path.join(
    os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'transformer-quantization_amd', 'harness',
    'weights.py'))
_hw = _ilu.module_from_spec(_spec)
_spec.loader.exec_module(_hw)
fill_from_numpy_stream, weight_check_sum = _hw.fill_from_numpy_stream, _hw.weight_check_sum
LAYERS, B, T = 2, 4, 64


def build_hf():
    torch.manual_seed(SEED)
    cfg = RobertaConfig(num_labels=2, num_hidden_layers=LAYERS, vocab_size=50265, max_position_embeddings=514,
                        type_vocab_size=1, pad_token_id=1)
    model = RobertaForSequenceClassification(cfg)
    model.eval()
    fill_from_numpy_stream(model, SEED)
    for layer in model.roberta.encoder.layer:       # transformers 4.1 semantics: functional GELU, folded by the reference
        del layer.intermediate.intermediate_act_fn
        object.__setattr__(layer.intermediate, 'intermediate_act_fn', torch.nn.functional.gelu)
    return model


def inputs():
    g = torch.Generator().manual_seed(SEED)
    ids = torch.randint(3, 50265, (B, T), generator=g)
    lengths = [T, T - 7, T // 2, 5]
    mask = torch.zeros(B, T, dtype=torch.long)
    for b, n in enumerate(lengths):
        mask[b, :n] = 1
        ids[b, n:] = 1                                # <pad>
    ids[:, 0] = 0                                     # <s>
    return ids, mask


def main():
    torch.set_num_threads(8)
    hf = build_hf()
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
              weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax,
              quant_dict={})
    emb = QuantizedRobertaEmbeddings(hf.roberta.embeddings, **qp)
    layers = [QuantizedRobertaLayer(l, **qp) for l in hf.roberta.encoder.layer]
    qp2 = dict(qp)
    qp2.pop('quant_dict')
    classifier = quantize_model(hf.classifier, **qp2)
    blocks = torch.nn.ModuleList([emb] + layers + [classifier])

    def apply(fn):
        for m in blocks.modules():
            if hasattr(m, fn) and not isinstance(m, QuantizationManager):
                getattr(m, fn)()

    pos_seen = []

    def forward(ids, attention_mask):
        mask = (1.0 - attention_mask[:, None, None, :].float()) * -10000.0      # HF get_extended_attention_mask (4.1)
        h = emb(input_ids=ids)
        for L in layers:
            att = L.attention
            ctx = att.self(h, mask)[0]
            a_out = att.output(ctx, h)
            h = L.output(L.intermediate(a_out), a_out)
        return classifier(h)

    # the position ids the reference derives (recorded for the harness test)
    from models.quantized_roberta import create_position_ids_from_input_ids
    blocks.eval()
    apply('quantized')
    ids, amask = inputs()
    pos_seen = create_position_ids_from_input_ids(ids, emb.padding_idx)
    with torch.no_grad():
        forward(ids, amask)                           # calibration batch (estimate_ranges state)
        for m in blocks.modules():
            if isinstance(m, QuantizationManager) and m.quantizer.is_initialized:
                m.fix_ranges()
        logits = forward(ids, amask)

    act, wts = [], []
    for name, m in blocks.named_modules():
        if isinstance(m, QuantizationManager) and m.quantizer.is_initialized:
            if name.endswith('activation_quantizer'):
                act.append((name, float(m.range_estimator.current_xmin), float(m.range_estimator.current_xmax)))
            elif name.endswith('weight_quantizer'):
                wts.append((name, float(m.quantizer._delta)))
    print('activation quantizers:', len(act), 'weight quantizers:', len(wts))
    print('logits', logits)
    np.savez_compressed(
        os.path.join(OUT, 'roberta_2l_w8a8.npz'),
        logits=logits.numpy(), input_ids=ids.numpy(), attention_mask=amask.numpy(), position_ids=pos_seen.numpy(),
        act_names=np.array([a[0] for a in act]), act_min=np.array([a[1] for a in act], np.float32),
        act_max=np.array([a[2] for a in act], np.float32),
        w_names=np.array([w[0] for w in wts]), w_delta=np.array([w[1] for w in wts], np.float32),
        versions=np.array(f'torch {torch.__version__} transformers {transformers.__version__}'),
        first_weight_sum=np.array(float(hf.roberta.encoder.layer[0].attention.self.query.weight.double().sum())),
        weight_check_sum=np.array(weight_check_sum(hf)))


if __name__ == '__main__':
    main()
