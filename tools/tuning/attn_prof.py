"""Phase profile of attention_i8_k: builds a -DTQ_ATTN_PROF copy of the library next to this file (s_memtime stamps,
csrc/tq_attention_i8.hip TQ_STAMP) and prints the per-phase cycle counts of the workgroups.

    python tools/tuning/attn_prof.py build      # here (CPU container): cross-compile the instrumented library
    python tools/tuning/attn_prof.py            # on the GPU box
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, 'transformer-quantization_amd')
OUT = os.path.join(ROOT, 'tools', 'tuning', '_prof')
LIB = os.path.join(OUT, 'libtq_hip.so')
sys.path.insert(0, PKG)
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == 'build':
    import build as B
    os.makedirs(OUT, exist_ok=True)
    srcs = B._sources()
    procs = [subprocess.Popen(['/opt/rocm/bin/hipcc'] + B.flags_for(s) + ['-DTQ_ATTN_PROF', '-c', s, '-o',
                                                                    os.path.join(OUT, os.path.basename(s)[:-4] + '.o')]) for s in srcs]
    assert all(p.wait() == 0 for p in procs)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] +
                          [os.path.join(OUT, os.path.basename(s)[:-4] + '.o') for s in srcs])
    print(LIB)
    sys.exit(0)

import torch
from quantization import _hip
_hip.LIB_PATH = LIB
be = _hip.backend()
p = lambda d, z: (torch.tensor(d).cuda(), torch.tensor(z).cuda(), None, 8, False, False, 1e-8)
names = ['V^T -> LDS', 'params + Q', 'scores MFMA', 'softmax/quant', 'barrier', 'PV + store']
for B_, T, H, dh in ((8, 128, 12, 64), (64, 128, 12, 64), (8, 128, 4, 32)):
    qi, ki, vi = (torch.randint(-128, 128, (B_, T, H * dh), dtype=torch.int8, device='cuda') for _ in range(3))
    mask = torch.zeros(B_, T, device='cuda')
    P = [p(0.02, 120.0), p(0.02, 130.0), p(0.01, 128.0), p(0.5, 128.0), p(0.003, 0.0), p(0.01, 128.0)]
    nblk = B_ * H * (T // 32)
    prof = torch.zeros(nblk * 8, dtype=torch.int64, device='cuda')
    os.environ['TQ_ATTN_PROF_PTR'] = hex(prof.data_ptr())
    for _ in range(3):
        be.attention_i8(qi, ki, vi, H, mask, float(dh) ** 0.5, *P, want_idx=True)
    torch.cuda.synchronize()
    t = prof.cpu().reshape(nblk, 8).double()
    d = (t[:, 1:7] - t[:, 0:6])
    print(f'B={B_} T={T} H={H} dh={dh}: {nblk} workgroups; s_memtime ticks (100 MHz => x10 ns), median / p90 over workgroups')
    for i, n in enumerate(names):
        print(f'   {n:14s} {d[:, i].median():8.0f} {d[:, i].quantile(0.9):8.0f}')
    print(f'   {"block total":14s} {(t[:, 6] - t[:, 0]).median():8.0f};  first start -> last end {(t[:, 6].max() - t[:, 0].min()):8.0f}')
