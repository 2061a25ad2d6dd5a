"""Which fixed-range forward is closer to the REFERENCE: the layered GPU route (hipBLASLt fp32 GEMMs + one kernel per
quantizer, the reference's module chain) or the exact-integer fused route (options.INT8_LINEAR + fused tails / attention)?

Judged against the reference's own outputs (tests/golden/bert_base_w8a8.npz, mobilebert_w4a4.npz: logits of the
reference's CPU forward), not against each other.  Two legs per model:

* `own_ranges`      -- the model is calibrated on the GPU (layered route, as always) and then evaluated both ways;
* `reference_ranges` -- the 161 / 774 activation ranges of the fixture are installed first (weights' grids are bit-equal
                         already), so the ONLY difference to the reference's forward is the arithmetic of the route.

Run on a GPU box: python scripts/int_vs_reference.py > gpurun_out/int_vs_reference.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'transformer-quantization_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

from quantization import options
from quantization.quantizers import QMethods
from quantization.range_estimators import RangeEstimators
from utils.utils import pass_data_for_range_estimation

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def census(model):
    from harness.bert import quantizer_census
    return quantizer_census(model)


def dev_stats(out, ref):
    d = np.abs(out.detach().float().cpu().numpy().astype(np.float64) - ref.astype(np.float64))
    span = float(ref.max() - ref.min())
    return {'max_abs': float(d.max()), 'mean_abs': float(d.mean()), 'max_over_span': float(d.max() / span),
            'mean_over_span': float(d.mean() / span), 'argmax_agree': float(
                (out.detach().float().cpu().numpy().argmax(-1) == ref.argmax(-1)).mean())}


def graph_ms(model, ids, n=30):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            model(ids)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        model(ids)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


class BertModes:
    def __init__(self):
        from harness.bert import QLayer, QResidualBlock, QSelfAttention
        self.c = (QLayer, QResidualBlock, QSelfAttention)

    def set(self, mode):
        QLayer, QResidualBlock, QSelfAttention = self.c
        options.INT8_LINEAR = mode == 'integer'
        QResidualBlock.fuse = mode in ('fused_tails', 'integer')
        QSelfAttention.fuse = mode in ('fused_tails', 'integer')
        QLayer.fuse_ffn = mode == 'integer'


class MobileModes:
    def __init__(self):
        from harness.mobilebert import QBottleneckLayer, QFFN, QMobileLayer, QMobileSelfAttention, QResidualNoNorm
        self.c = (QBottleneckLayer, QFFN, QMobileSelfAttention, QResidualNoNorm)
        self.L = QMobileLayer

    def set(self, mode):
        options.INT8_LINEAR = mode == 'integer'
        for c in self.c:
            c.fuse = mode == 'integer'
        self.L.fuse_ffn = mode == 'integer'
        if mode == 'fused_tails':
            self.c[3].fuse = True


def install_reference_ranges(act, ref_ranges):
    """ref_ranges: list of (xmin, xmax) in census order."""
    for (_, m), (lo, hi) in zip(act, ref_ranges):
        dev = m.quantizer._delta.device
        lo_t = torch.tensor(float(lo), dtype=torch.float32, device=dev)
        hi_t = torch.tensor(float(hi), dtype=torch.float32, device=dev)
        m.quantizer.set_quant_range(lo_t, hi_t)
        m.range_estimator.current_xmin = lo_t
        m.range_estimator.current_xmax = hi_t


def hidden_stats(model, ids, zh):
    """Encoder output after layers 1 / 6 / 12 against the reference's (tests/golden/bert_base_w8a8_hidden.npz), in steps of
    the REFERENCE's grid at that site: 786 432 samples per layer instead of 16 logits."""
    got = {}
    hooks = [model.layers[k - 1].register_forward_hook(lambda m, i, o, k=k: got.__setitem__(k, o.detach()))
             for k in (1, 6, 12)]
    try:
        model(ids.cuda())
    finally:
        for h in hooks:
            h.remove()
    r = {}
    for k in (1, 6, 12):
        d = float(zh[f'hidden_delta_L{k}'])
        zp = float(np.clip(np.rint(zh[f'hidden_zero_float_L{k}']), 0, 255))
        ref = (zh[f'hidden_idx_L{k}'].astype(np.float64) - zp) * d
        dev = np.abs(got[k].double().cpu().numpy() - ref) / d
        r[f'L{k}'] = {'same_grid_point_frac': float((dev < 0.5).mean()), 'mean_abs_dev_steps': float(dev.mean()),
                      'max_abs_dev_steps': float(dev.max())}
    return r


def run_model(name, model, ids, ref_logits, ref_ranges, modes, out, zh=None):
    res = {}
    with torch.no_grad():
        pass_data_for_range_estimation([(ids,)], model, act_quant=True, weight_quant=True, max_num_batches=1)
        model.fix_ranges()
        act, wts = census(model)
        for leg in ('own_ranges', 'reference_ranges'):
            if leg == 'reference_ranges':
                install_reference_ranges(act, ref_ranges)
                options.invalidate_derived_caches()
            r = {}
            for mode in ('layered', 'fused_tails', 'integer'):
                modes.set(mode)
                try:
                    o = model(ids.cuda())
                    r[mode] = dev_stats(o, ref_logits)
                    r[mode]['deterministic'] = bool(torch.equal(o, model(ids.cuda())))
                    if zh is not None:
                        r[mode]['hidden_vs_reference'] = hidden_stats(model, ids, zh)
                        if leg == 'reference_ranges':
                            ex = torch.from_numpy(zh['input_ids_extra'])
                            lo = torch.cat([o] + [model(ex[i].cuda()) for i in range(ex.shape[0])])
                            r[mode]['logits_4_batches'] = dev_stats(lo, np.concatenate([ref_logits] + list(zh['logits_extra'])))
                    if leg == 'own_ranges':
                        r[mode]['hipgraph_ms'] = round(graph_ms(model, ids.cuda()), 4)
                finally:
                    modes.set('layered')
            r['integer_closer_or_equal'] = bool(r['integer']['max_abs'] <= r['layered']['max_abs'] * 1.1)
            res[leg] = r
    res['logit_span_reference'] = float(ref_logits.max() - ref_logits.min())
    out[name] = res


def main():
    out = {'what': __doc__.split('\n\n')[0]}
    # ---- BERT-base W8A8 (BASELINE configs[0]/[1]) ------------------------------------------------------------------
    from harness.bert import build_bert_base
    z = np.load(os.path.join(GOLDEN, 'bert_base_w8a8.npz'))
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
              weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)
    model, _ = build_bert_base(seed=1000, **qp)
    model = model.cuda().eval()
    ids = torch.from_numpy(z['input_ids'])
    zh = np.load(os.path.join(GOLDEN, 'bert_base_w8a8_hidden.npz'))
    run_model('bert_base_w8a8', model, ids, z['logits'], list(zip(z['act_min'], z['act_max'])), BertModes(), out, zh)
    del model
    # ---- MobileBERT W4A4 (config 5's model, fixture of config 4) ----------------------------------------------------
    from harness.mobilebert import build_mobilebert
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from tests.test_mobilebert_e2e import _ref_name
    z = np.load(os.path.join(GOLDEN, 'mobilebert_w4a4.npz'))
    for bits, key in ((4, 'mobilebert_w4a4'),):
        qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=bits, n_bits_act=bits,
                  weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax,
                  quant_dict={'attn_probs_n_bits_act': 8})
        model, _ = build_mobilebert(seed=1000, **qp)
        model = model.cuda().eval()
        ids = torch.from_numpy(z['input_ids'])
        names = [str(n) for n in z['act_names']]
        by = {n: (z['act_min'][i], z['act_max'][i]) for i, n in enumerate(names)}
        act, _ = census(model)
        ref_ranges = [by[_ref_name(n)] for n, _ in act]
        run_model(key, model, ids, z['logits'], ref_ranges, MobileModes(), out)
        del model
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
