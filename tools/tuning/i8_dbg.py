"""Where the time of the fused integer Linear goes at M = 8192 (768 -> 3072, GELU + quantizer, index-only):
TQ_I8_DBG bit 1 = no epilogue, 2 = no operand loads, 4 = no MFMA (tq_linear_i8.hip, read per call).
Needs a breakdown build:  TQ_EXTRA_HIPCC_FLAGS=-DTQ_I8_DBG_BUILD python transformer-quantization_amd/build.py --force"""
import os, sys, time
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization import _hip
be = _hip.backend(); dev = 'cuda'
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8192, 3072, 768)))
x = torch.randint(-128, 127, (M, K), dtype=torch.int8, device=dev); w = torch.randint(-127, 127, (N, K), dtype=torch.int8, device=dev)
rs = be.rowsum_i8(w); b = torch.randn(N, device=dev)
xd = torch.tensor(0.02, device=dev); xz = torch.tensor(117.0, device=dev); wd = torch.tensor(0.001, device=dev).reshape(1)
qo = (torch.tensor(0.05, device=dev), torch.tensor(100.0, device=dev), None, 8, False, False, 1e-8)
def run(want_y):
    return be.linear_i8(x, w, rs, b, (xd, xz, 8, 1e-8), wd, 1e-8, _hip.ACT_GELU, qo, torch.float32, want_idx=True, want_y=want_y)
def t(fn, n=30):
    a = torch.empty(1 << 26, device=dev); te = time.perf_counter() + 0.3
    while time.perf_counter() < te:
        a.add_(1.0); torch.cuda.synchronize()
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
print('LDS pad', os.environ.get('TQ_I8_LDS_PAD', '0'))
for dbg, what in ((0, 'full kernel'), (1, 'no epilogue'), (3, 'no epilogue, no loads'), (5, 'no epilogue, no MFMA'), (7, 'barriers + LDS reads only'),
                  (4, 'epilogue + loads, no MFMA'), (2, 'no loads (MFMA + epilogue)')):
    os.environ['TQ_I8_DBG'] = str(dbg)
    print(f'dbg={dbg} {what:32s} index-only {t(lambda: run(False)):7.1f} us   with fp32 y {t(lambda: run(True)):7.1f} us', flush=True)
