"""tq_linear_i8_fwd at calibration-sized shapes (M up to 16384 tokens): with / without the output quantizer, activation, fp32 y."""
import sys, time
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization import _hip
be = _hip.backend()
dev = 'cuda'
# settle the clocks
z = torch.zeros(1 << 26, device=dev)
t = time.perf_counter() + 0.5
while time.perf_counter() < t:
    for _ in range(10): z.add_(1.0)
    torch.cuda.synchronize()
for M, N, K in ((8192, 3072, 768), (16384, 3072, 768), (16384, 768, 3072), (16384, 768, 768), (4096, 3072, 768)):
    x = torch.randint(-128, 127, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(-127, 127, (N, K), dtype=torch.int8, device=dev)
    rs = be.rowsum_i8(w)
    bias = torch.randn(N, device=dev)
    xq = (torch.tensor(0.02, device=dev), torch.tensor(117.0, device=dev), 8, 1e-8)
    wd = torch.tensor(0.001, device=dev).reshape(1)
    qo = (torch.tensor(0.05, device=dev), torch.tensor(100.0, device=dev), None, 8, False, False, 1e-8)
    for name, act, q, want_idx, want_y in (('gelu + Q, y', 2, qo, False, True), ('gelu + Q, y + idx', 2, qo, True, True), ('gelu + Q, idx only', 2, qo, True, False),
                                           ('gelu, y (no Q)', 2, None, False, True), ('no act, y (no Q)', 0, None, False, True)):
        f = lambda: be.linear_i8(x, w, rs, bias, xq, wd, 1e-8, act, q, torch.float32, want_idx=want_idx, want_y=want_y)
        for _ in range(10): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        print(f'M={M} N={N} K={K} {name:20s}: {us:8.1f} us  {2 * M * N * K / us / 1e6:7.1f} TOP/s')
