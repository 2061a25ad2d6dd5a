"""Which fixed-range forward is closer to the REFERENCE: the layered GPU route (hipBLASLt fp32 GEMMs + one kernel per
quantizer, the reference's module chain) or the exact-integer fused route (options.INT8_LINEAR + fused tails / attention)?

Judged against outputs of the reference itself -- tests/golden/{bert_base_w8a8,mobilebert_w4a4}_hidden.npz: the grid
indices of the encoder output after selected layers (0.5-0.8 M samples each) and the logits of four evaluation batches --
not against each other.  Two legs per model:

* `own_ranges`       -- the model is calibrated on the GPU (layered route, as always) and then evaluated each way;
* `reference_ranges` -- the 161 / 774 activation ranges of the reference are installed first (the weights' grids are
                        bit-equal already), so the ONLY difference to the reference's forward is the route's arithmetic.

Run on a GPU box: python scripts/int_vs_reference.py > gpurun_out/r05/int_vs_reference.json
(the same comparison is asserted in tests/test_bert_e2e.py / tests/test_mobilebert_e2e.py `..._default_route_...`).
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

from harness.routes import Route, compare_routes, install_reference_ranges
from quantization.quantizers import QMethods
from quantization.range_estimators import RangeEstimators
from utils.utils import pass_data_for_range_estimation

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
ROUTES = ('layered', 'fused_tails', 'integer', 'default')


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


def run_model(model, ids, zh, layers, ref_ranges):
    from harness.bert import quantizer_census
    res = {}
    with torch.no_grad():
        pass_data_for_range_estimation([(ids,)], model, act_quant=True, weight_quant=True, max_num_batches=1)
        model.fix_ranges()
        act, _ = quantizer_census(model)
        for leg in ('own_ranges', 'reference_ranges'):
            if leg == 'reference_ranges':
                install_reference_ranges([m for _, m in act], ref_ranges(act))
            r = compare_routes(model, ids, zh, layers, routes=ROUTES)
            for name, v in r.items():
                lo = v.pop('logits')
                v['deterministic'] = True
                if leg == 'own_ranges':
                    with Route(model, name):
                        v['deterministic'] = bool(torch.equal(lo[:ids.shape[0]], model(ids.cuda())))
                        v['hipgraph_ms'] = round(graph_ms(model, ids.cuda()), 4)
            r['default_equals_integer'] = bool(r['default']['logits_4_batches'] == r['integer']['logits_4_batches'])
            res[leg] = r
    return res


def main():
    out = {'what': __doc__.split('\n\n')[0]}
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
              weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)
    # ---- BERT-base W8A8 (BASELINE configs[0] / [1]) ---------------------------------------------------------------------
    from harness.bert import build_bert_base
    z = np.load(os.path.join(GOLDEN, 'bert_base_w8a8.npz'))
    zh = np.load(os.path.join(GOLDEN, 'bert_base_w8a8_hidden.npz'))
    model, _ = build_bert_base(seed=1000, **qp)
    model = model.cuda().eval()
    out['bert_base_w8a8'] = run_model(model, torch.from_numpy(z['input_ids']), zh, (1, 6, 12),
                                      lambda act: list(zip(z['act_min'], z['act_max'])))
    del model
    # ---- MobileBERT W4A4 (config 5's model) ------------------------------------------------------------------------------
    from harness.mobilebert import build_mobilebert
    from tests.test_mobilebert_e2e import _ref_name
    z = np.load(os.path.join(GOLDEN, 'mobilebert_w4a4.npz'))
    zh = np.load(os.path.join(GOLDEN, 'mobilebert_w4a4_hidden.npz'))
    qp4 = dict(qp, n_bits=4, n_bits_act=4, quant_dict={'attn_probs_n_bits_act': 8})
    model, _ = build_mobilebert(seed=1000, **qp4)
    model = model.cuda().eval()
    by = {str(n): (z['act_min'][i], z['act_max'][i]) for i, n in enumerate(z['act_names'])}
    out['mobilebert_w4a4'] = run_model(model, torch.from_numpy(z['input_ids']), zh, (1, 6, 12, 24),
                                       lambda act: [by[_ref_name(n)] for n, _ in act])
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
