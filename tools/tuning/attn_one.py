"""A few launches of tq_attention_i8_fwd at one shape (for rocprofv3 --pmc passes, scripts/pmc_attention.sh).

    python tools/tuning/attn_one.py B [T H dh]"""
import sys
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization import _hip
be = _hip.backend()
a = [int(v) for v in sys.argv[1:]]
B = a[0] if a else 64
T, H, dh = (a[1:4] if len(a) >= 4 else (128, 12, 64))
p = lambda d, z: (torch.tensor(d).cuda(), torch.tensor(z).cuda(), None, 8, False, False, 1e-8)
torch.manual_seed(0)
qi, ki, vi = (torch.randint(-128, 128, (B, T, H * dh), dtype=torch.int8, device='cuda') for _ in range(3))
mask = torch.zeros(B, T, device='cuda')
P = [p(0.02, 120.0), p(0.02, 130.0), p(0.01, 128.0), p(0.5, 128.0), p(0.003, 0.0), p(0.01, 128.0)]
for _ in range(10):
    be.attention_i8(qi, ki, vi, H, mask, float(dh) ** 0.5, *P, want_idx=True)
torch.cuda.synchronize()
