"""Kernel durations of the MobileBERT feed-forward chain (tq_ffn_chain_i8_nonorm_fwd) for 2..4 blocks next to the
single-block kernel, M = 1024:   rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o t -- python tools/tuning/ffn_chain_time.py"""
import sys
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization import _hip
be = _hip.backend(); dev = 'cuda'
M, K1, N1, N2 = 1024, 128, 512, 128
g = torch.Generator().manual_seed(0)
q7 = lambda d, z: (torch.tensor(d, device=dev), torch.tensor(z, device=dev), None, 8, False, False, 1e-8)
x_i8 = torch.randint(-128, 127, (M, K1), dtype=torch.int8, device=dev)
xq = (torch.tensor(0.02, device=dev), torch.tensor(128.0, device=dev), 8, 1e-8)
res = torch.randn(M, N2, device=dev)
stages = []
for k in range(4):
    w1 = torch.randint(-127, 127, (N1, K1), dtype=torch.int8, device=dev); w2 = torch.randint(-127, 127, (N2, N1), dtype=torch.int8, device=dev)
    stages.append(dict(w1_idx=w1, w1_rowsum=be.rowsum_i8(w1), bias1=torch.randn(N1, device=dev), w1_delta=torch.tensor([0.001], device=dev),
                       w1_eps=1e-8, q_mid=q7(0.01, 0.0), w2_idx=w2, w2_rowsum=be.rowsum_i8(w2), bias2=torch.randn(N2, device=dev),
                       w2_delta=torch.tensor([0.001], device=dev), w2_eps=1e-8, nn_w=torch.randn(N2, device=dev), nn_b=torch.randn(N2, device=dev),
                       q_dense=q7(0.02, 128.0), q_sum=q7(0.03, 128.0), q_out=q7(0.02, 128.0)))
s0 = stages[0]
for _ in range(30):
    be.ffn_i8_nonorm(x_i8, xq, s0['w1_idx'], s0['w1_rowsum'], s0['bias1'], s0['w1_delta'], 1e-8, s0['q_mid'], s0['w2_idx'], s0['w2_rowsum'],
                     s0['bias2'], s0['w2_delta'], 1e-8, res, s0['nn_w'], s0['nn_b'], s0['q_dense'], s0['q_sum'], s0['q_out'], torch.float32, want_idx=True)
for n in (2, 3, 4):
    for _ in range(30):
        be.ffn_chain_i8_nonorm(x_i8, xq, res, stages[:n], torch.float32, want_idx=True)
torch.cuda.synchronize()
