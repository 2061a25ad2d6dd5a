"""Fused residual tails (LayerNorm and NoNorm) at HBM-bound sizes."""
import sys
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization import _hip
be = _hip.backend()
def ev(fn, n=20, w=5, rounds=3):
    for _ in range(w): fn()
    best = 1e9
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n)
    return best
q = lambda d_, z_: (torch.tensor(d_, device='cuda'), torch.tensor(z_, device='cuda'), None, 8, False, False, 1e-8)
for rows, d in ((131072, 768), (131072, 512), (131072, 3072), (524288, 128), (1024, 768)):
    for dt in (torch.float32, torch.bfloat16):
        a = torch.randn(rows, d, device='cuda').to(dt); r = torch.randn(rows, d, device='cuda').to(dt)
        w = torch.randn(d, device='cuda'); b = torch.randn(d, device='cuda')
        q1, q2, q3 = q(0.05, 120.0), q(0.06, 128.0), q(0.03, 128.0)
        for name, eps in (('LayerNorm', 1e-12), ('NoNorm', None)):
            ms = ev(lambda: be.residual_layernorm_quant(a, r, q1, q2, w, b, eps, q3))
            print(f'[{rows},{d}] {str(dt)[6:]:9s} {name:9s} {ms*1e3:8.1f} us  {3*a.numel()*a.element_size()/ms/1e9:6.2f} TB/s')
