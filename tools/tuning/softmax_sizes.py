import sys, time
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization import _hip
be = _hip.backend(); dev = 'cuda'
def q7(d, z): return (torch.tensor(d, device=dev), torch.tensor(z, device=dev), None, 8, False, False, 1e-8)
qs, qp = q7(0.5, 128.0), q7(0.004, 0.0)
for B in (8, 64, 256, 512):
    s = torch.randn(B, 12, 128, 128, device=dev); mask = torch.zeros(B, 128, device=dev)
    a = torch.empty(1 << 26, device=dev); te = time.perf_counter() + 0.3
    while time.perf_counter() < te:
        a.add_(1.0); torch.cuda.synchronize()
    for _ in range(3): be.scores_softmax_quant(s, mask, 12 * 128, 8.0, qs, qp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): be.scores_softmax_quant(s, mask, 12 * 128, 8.0, qs, qp)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print(f'B={B:4d} {s.numel() * 8 / 1e6:8.1f} MB  {us:8.1f} us  {s.numel() * 8 / us / 1e3:7.1f} GB/s  {s.numel() * 8 / us / 1e3 / 80:5.1f} %')
