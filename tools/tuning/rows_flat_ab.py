"""Per-row quantizers (per-token activations, per-channel weights): the flat-tile kernel fq_rows_flat against the
wave-per-row kernels (TQ_ROWS_FLAT=0), HIP events over 30 launches, one process per setting on the same box.
    python tools/tuning/rows_flat_ab.py            # runs both settings"""
import os, subprocess, sys
if len(sys.argv) == 1:
    for flat in ('0', '1'):
        print(f'== TQ_ROWS_FLAT={flat}', flush=True)
        subprocess.check_call([sys.executable, __file__, 'run'], env=dict(os.environ, TQ_ROWS_FLAT=flat))
    sys.exit(0)
sys.path.insert(0, '/root/repo/transformer-quantization_amd')
import torch
from quantization import _hip
be = _hip.backend() if hasattr(_hip, 'backend') else _hip
dev = 'cuda'


def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


SHAPES = (((1024, 512, 768), 512, 768), ((512, 512, 768), 512, 768), ((256, 512, 768), 512, 768), ((128, 512, 768), 512, 768),
          ((64, 512, 768), 512, 768), ((64, 128, 3072), 128, 3072), ((8, 128, 768), 128, 768),
          ((8, 128, 3072), 128, 3072), ((3072, 768), 3072, 768), ((30522, 768), 30522, 768))
for shape, axis_n, inner in SHAPES:
    for dt, es in ((torch.bfloat16, 2), (torch.float32, 4)):
        x = torch.randn(*shape, device=dev).to(dt)
        n = x.numel()
        d = torch.rand(axis_n, device=dev) * 0.02 + 0.02
        z = torch.full((axis_n,), 128.0, device=dev)
        us = timeit(lambda: be.fake_quant(x, d, z, None, 8, False, False, 1e-8, axis_n, inner))
        us_i = timeit(lambda: be.fake_quant(x, d, z, None, 8, False, False, 1e-8, axis_n, inner, want_y=False, idx_dtype=torch.uint8))
        print(f'{str(shape):18s} {str(dt)[6:]:9s} y: {us:8.1f} us {2 * es * n / us / 1e3:7.0f} GB/s {2 * es * n / us / 8e4:5.1f} %   '
              f'index-only u8: {us_i:8.1f} us {(es + 1) * n / us_i / 8e4:5.1f} %', flush=True)
        del x
