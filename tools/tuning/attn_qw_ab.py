"""A/B of the query waves per workgroup of tq_attention_i8_fwd (TQ_ATTN_QW = 2 / 8, read per call) as graph replays."""
import os, sys, time
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization import _hip
be = _hip.backend()
p = lambda d, z: (torch.tensor(d).cuda(), torch.tensor(z).cuda(), None, 8, False, False, 1e-8)
N = 40
for B, T, H, dh in ((16, 128, 12, 64), (32, 128, 12, 64), (64, 128, 12, 64), (128, 128, 12, 64), (32, 256, 12, 64), (16, 512, 12, 64), (64, 128, 4, 32)):
    qi, ki, vi = (torch.randint(-128, 128, (B, T, H * dh), dtype=torch.int8, device='cuda') for _ in range(3))
    mask = torch.zeros(B, T, device='cuda')
    P = [p(0.02, 120.0), p(0.02, 130.0), p(0.01, 128.0), p(0.5, 128.0), p(0.003, 0.0), p(0.01, 128.0)]
    f = lambda: be.attention_i8(qi, ki, vi, H, mask, float(dh) ** 0.5, *P, want_idx=True)
    res = {}
    for qw in ('2', '8', 'auto'):
        if qw == 'auto':
            os.environ.pop('TQ_ATTN_QW', None)
        else:
            os.environ['TQ_ATTN_QW'] = qw
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): f()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(N): f()
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        res[qw] = (time.perf_counter() - t0) / 20 / N * 1e6
    print(f'B={B} T={T} H={H} dh={dh}: 2 waves {res["2"]:7.2f} us   8 waves {res["8"]:7.2f} us   auto {res["auto"]:7.2f} us')
