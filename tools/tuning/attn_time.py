"""tq_attention_i8_fwd timing at BERT-base shapes."""
import sys
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization import _hip
be = _hip.backend()
def ev(fn, n=50, w=10, rounds=3):
    for _ in range(w): fn()
    best = 1e9
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n)
    return best * 1e3
p = lambda d, z: (torch.tensor(d).cuda(), torch.tensor(z).cuda(), None, 8, False, False, 1e-8)
for B, T, H in ((8, 128, 12), (32, 128, 12), (8, 256, 12), (8, 64, 12), (64, 128, 16)):
    qi, ki, vi = (torch.randint(-128, 128, (B, T, H * 64), dtype=torch.int8, device='cuda') for _ in range(3))
    mask = torch.zeros(B, T, device='cuda')
    P = [p(0.02, 120.0), p(0.02, 130.0), p(0.01, 128.0), p(0.5, 128.0), p(0.003, 0.0), p(0.01, 128.0)]
    f = lambda: be.attention_i8(qi, ki, vi, H, mask, 8.0, *P, want_idx=True)
    us = ev(f)
    flops = 2 * 2 * B * H * T * T * 64
    print(f'B={B} T={T} H={H}: {us:8.1f} us   {flops/us/1e6:8.1f} TOP/s (int8 MACs x2)')
