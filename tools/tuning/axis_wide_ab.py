"""Per-embedding quantizers on wide rows (the [B, T, 3072] FFN activations): LDS-table kernel fq_axis against the
register kernel fq_axis_reg (TQ_AXIS_REG = 0 / 1; 2 = the shipped choice), HIP events, one process per setting."""
import os, subprocess, sys
if len(sys.argv) == 1:
    for v in ('0', '2'):
        print(f'== TQ_AXIS_REG={v}', flush=True)
        subprocess.check_call([sys.executable, __file__, 'run'], env=dict(os.environ, TQ_AXIS_REG=v))
    sys.exit(0)
if sys.argv[1] == 'tpb':                     # register kernel, tiles per block
    for t in ('4', '8', '16', '32'):
        print(f'== TQ_AXIS_REG=1 TQ_AXIS_REG_TPB={t}', flush=True)
        subprocess.check_call([sys.executable, __file__, 'run', 'big'], env=dict(os.environ, TQ_AXIS_REG='1', TQ_AXIS_REG_TPB=t))
    sys.exit(0)
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


BIG = ((256, 128, 3072), (1024, 128, 3072), (1024, 512, 768), (256, 512, 768))
for shape in BIG if len(sys.argv) > 2 else ((256, 512, 768), (8, 128, 3072), (64, 128, 3072), (256, 128, 3072), (1024, 128, 3072), (8, 128, 768), (64, 128, 768), (1024, 512, 768),
              (64, 128, 512), (64, 128, 1024), (64, 128, 4096)):
    D = shape[-1]
    for dt, es in ((torch.bfloat16, 2), (torch.float32, 4)):
        x = torch.randn(*shape, device=dev).to(dt)
        n = x.numel()
        dv, zv = torch.rand(D, device=dev) * 0.02 + 0.02, torch.full((D,), 128.0, device=dev)
        us = timeit(lambda: be.fake_quant(x, dv, zv, None, 8, False, False, 1e-8, D, 1))
        us_i = timeit(lambda: be.fake_quant(x, dv, zv, None, 8, False, False, 1e-8, D, 1, want_y=False, idx_dtype=torch.uint8))
        print(f'{str(shape):18s} {str(dt)[6:]:9s} y: {us:8.1f} us {2 * es * n / us / 8e4:5.1f} %   index-only u8: {us_i:8.1f} us {(es + 1) * n / us_i / 8e4:5.1f} %', flush=True)
        del x
