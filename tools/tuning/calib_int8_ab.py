"""BERT-base calibrating forward (running min-max, W8A8) with options.INT8_CALIBRATION on / off: eager and hipGraph times at
[8,128] and [128,128], the deviation between the ranges the two routes estimate, and how many Linears ran as integer GEMMs."""
import sys, time
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from harness.bert import build_bert_base, quantizer_census
from quantization import options
from quantization.autoquant_utils import INT8_STATS
from quantization.graphs import GraphedForward
from quantization.quantizers import QMethods
from quantization.range_estimators import RangeEstimators

qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
          weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)


def wall(f, n=10, w=2):
    for _ in range(w): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ranges = {}
for B in (8, 32, 128):
    ids = torch.randint(1000, 30000, (B, 128), device='cuda', generator=torch.Generator(device='cuda').manual_seed(4000))
    for on in (False, True):
        options.INT8_CALIBRATION = on
        model, _ = build_bert_base(seed=1000, **qp)
        model = model.cuda().eval()
        with torch.no_grad():
            model.set_quant_state(True, True)
            model.estimate_ranges()
            before = INT8_STATS['kernel_calls']
            model(ids)
            calls = INT8_STATS['kernel_calls'] - before
            act, _ = quantizer_census(model)
            ranges[B, on] = torch.stack([torch.stack([m.quantizer.x_min.reshape(()), m.quantizer.x_max.reshape(())]) for _, m in act]).cpu()
            eager = wall(lambda: model(ids))
            options.INPLACE_CALIBRATION_STATE = True
            try:
                model(ids)
                gf = GraphedForward(model, ids)
                graph = wall(lambda: gf(ids), 20, 3)
                del gf
            finally:
                options.INPLACE_CALIBRATION_STATE = False
        print(f'[{B},128] INT8_CALIBRATION={on}: integer Linears in the first calibrating forward {calls}; eager {eager:.3f} ms, hipGraph {graph:.3f} ms '
              f'({B * 128 / graph * 1e3:.0f} tokens/s)')
    a, b = ranges[B, False], ranges[B, True]
    span = (a[:, 1] - a[:, 0]).abs().clamp_min(1e-12)
    dev = ((a - b).abs().max(dim=1).values / span)
    print(f'[{B},128] ranges after one batch, integer vs layered calibration: {len(dev)} sites, max deviation {float(dev.max()):.2e} of the site span, median {float(dev.median()):.2e}')
