"""Index outputs (uint8 / int8(index - 128)) with and without the dequantised tensor, HIP events over 30 launches; run once per
library (TQ_LIB_PATH) on the same box:  python tools/tuning/idx_only_time.py"""
import sys
sys.path.insert(0, '/root/repo/transformer-quantization_amd')
import torch
from quantization import _hip
be = _hip.backend()
dev = 'cuda'


def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


K1 = len(sys.argv) > 1 and sys.argv[1] == 'k1'          # only the headline shape, with the plain fake-quant row
for shape in ((1024, 512, 768),) if K1 else ((1024, 512, 768), (64, 128, 3072), (8, 128, 768)):
    D = shape[-1]
    for dt, es in ((torch.bfloat16, 2), (torch.float32, 4)):
        x = torch.randn(*shape, device=dev).to(dt)
        n = x.numel()
        d1, z1 = torch.tensor(0.03, device=dev), torch.tensor(128.0, device=dev)
        dv, zv = torch.full((D,), 0.03, device=dev), torch.full((D,), 128.0, device=dev)
        rows = (('per-tensor y (K1)', lambda: be.fake_quant(x, d1, z1, None, 8, False, False, 1e-8, 1, 1), 2 * es),
                ('per-tensor index-only u8', lambda: be.fake_quant(x, d1, z1, None, 8, False, False, 1e-8, 1, 1, want_y=False, idx_dtype=torch.uint8), es + 1),
                ('per-tensor y + int8(idx-128)', lambda: be.fake_quant_int8(x, d1, z1, 8, 1e-8), 2 * es + 1),
                ('per-embedding index-only u8', lambda: be.fake_quant(x, dv, zv, None, 8, False, False, 1e-8, D, 1, want_y=False, idx_dtype=torch.uint8), es + 1))
        for name, fn, bpe in rows:
            us = timeit(fn)
            print(f'{str(shape):18s} {str(dt)[6:]:9s} {name:30s} {us:8.1f} us {bpe * n / us / 1e3:7.0f} GB/s {bpe * n / us / 8e4:5.1f} %', flush=True)
        del x
