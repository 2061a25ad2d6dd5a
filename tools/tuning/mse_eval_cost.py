"""What does ONE golden-section loss evaluation cost (README recipe: ~20 per weight tensor, 102 tensors)?
device time of the launches (no sync), the same with the device->host copy of the loss, and the whole loss_fx call."""
import sys, time
sys.path[:0] = ['/root/repo/transformer-quantization_amd', '/root/repo']
import numpy as np, torch
from quantization import _hip
from quantization.quantizers import QMethods
from quantization.range_estimators import RangeEstimators, OptMethod, candidate_params
be = _hip.backend()
for shape in ((768, 768), (3072, 768), (768, 3072), (768,), (30522, 768)):
    w = torch.randn(*shape, device='cuda') * 0.05
    cand = be.candidate_table(candidate_params([-0.2], [0.2], 8, True), w.device)
    loss = be.zeros_f64((1, 1), w.device)
    for _ in range(5):
        be.mse_candidates_ordered(w, cand, loss)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        be.mse_candidates_ordered(w, cand, loss)
    torch.cuda.synchronize()
    t_async = (time.perf_counter() - t0) / n * 1e6
    t0 = time.perf_counter()
    for _ in range(n):
        be.mse_candidates_ordered(w, cand, loss)
        loss.cpu()
    t_sync = (time.perf_counter() - t0) / n * 1e6
    est = RangeEstimators.MSE.cls(quantizer=QMethods.symmetric_uniform.cls(n_bits=8), opt_method=OptMethod.golden_section)
    est(w)
    t0 = time.perf_counter()
    for _ in range(n):
        est.loss_fx(w, -0.2, 0.2)
    t_fx = (time.perf_counter() - t0) / n * 1e6
    print(f'{str(shape):14s} launches back to back {t_async:7.1f} us | + device->host copy of the loss {t_sync:7.1f} us | loss_fx {t_fx:7.1f} us')
