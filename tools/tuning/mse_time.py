"""MSE search timings (whole QuantizationManager.forward incl. table build) + raw candidate kernel."""
import sys, time
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch, numpy as np
from quantization import _hip
from quantization.quantization_manager import QuantizationManager
from quantization.quantizers import QMethods
from quantization.range_estimators import RangeEstimators, OptMethod
be = _hip.backend()
def wall(fn, n=5, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
x = torch.randn(8, 128, 768, device='cuda'); x[..., 308] *= 20
for name, method, params in (('1d_sym8_100', 'symmetric_uniform', dict(num_candidates=100)),
                             ('2d_asym8_100x64x2', 'asymmetric_uniform', dict(num_candidates=100)),
                             ('golden_sym8', 'symmetric_uniform', dict(opt_method=OptMethod.golden_section))):
    def one():
        mgr = QuantizationManager(qmethod=QMethods[method], init=RangeEstimators.MSE, qparams=dict(n_bits=8), init_params=params)
        mgr(x)
    print(name, '[8,128,768] ms', wall(one))
for shape in ((8, 128, 768), (256, 512, 768)):
    xx = torch.randn(*shape, device='cuda')
    for C in (128, 12800):
        tab = torch.tensor(np.stack([np.linspace(0.01, 0.2, C), np.full(C, 100.0), np.zeros(C), np.full(C, 255.0)], 1).astype(np.float32)).cuda()
        loss = be.zeros_f64((1, C), 'cuda')
        ms = wall(lambda: be.mse_candidates(xx, 1, tab, loss), n=5, w=1)
        print(f'raw kernel {shape} C={C}: {ms:.3f} ms  {xx.numel()*C/ms/1e9:.1f} T cand-elem/s')
# ordered (reference summation order) kernel, same shapes
import os
for shape in ((8, 128, 768), (3072, 768), (256, 512, 768)):
    xx = torch.randn(*shape, device='cuda')
    for C in (1, 100, 12800):
        if C == 12800 and shape[0] == 256:
            continue
        tab = torch.tensor(np.stack([np.linspace(0.01, 0.2, C), np.full(C, 100.0), np.zeros(C), np.full(C, 255.0)], 1).astype(np.float32)).cuda()
        loss = be.zeros_f64((1, C), 'cuda')
        for kt in (0, 2, 3, 4):
            os.environ['TQ_ORD_KTOP'] = str(kt)
            ms = wall(lambda: be.mse_candidates_ordered(xx, tab, loss), n=5, w=1)
            print(f'ordered kernel {shape} C={C} ktop={kt}: {ms:.3f} ms  {xx.numel()*C/ms/1e9:.2f} T cand-elem/s')
        os.environ.pop('TQ_ORD_KTOP')
