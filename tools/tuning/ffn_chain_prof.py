"""Phase profile of ffn_chain_i8_k (s_memtime stamps, csrc/tq_linear_i8.hip TQ_FSTAMP): build a -DTQ_FFN_PROF copy of the
library next to this file and print the per-phase cycle counts per stage.
    python tools/tuning/ffn_chain_prof.py build      # here: cross-compile the instrumented library
    python tools/tuning/ffn_chain_prof.py            # on the GPU box"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, 'transformer-quantization_amd'); OUT = os.path.join(ROOT, 'tools', 'tuning', '_prof')
LIB = os.path.join(OUT, 'libtq_hip_ffn.so')
sys.path.insert(0, PKG); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == 'build':
    import build as B
    os.makedirs(OUT, exist_ok=True)
    obj = os.path.join(OUT, 'tq_linear_i8_ffnprof.o')
    subprocess.check_call(['/opt/rocm/bin/hipcc'] + B.FLAGS + ['-DTQ_FFN_PROF', '-c', os.path.join(B.CSRC, 'tq_linear_i8.hip'), '-o', obj])
    others = [os.path.join(B.OBJDIR, f) for f in os.listdir(B.OBJDIR) if f.endswith('.o') and f != 'tq_linear_i8.o' and 'dbg' not in f]
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, obj] + others)
    print(LIB); sys.exit(0)
import torch
from quantization import _hip
_hip.LIB_PATH = LIB
sys.argv = sys.argv[:1]
import importlib.util
spec = importlib.util.spec_from_file_location('fct', os.path.join(ROOT, 'tools', 'tuning', 'ffn_chain_time.py'))
prof = torch.zeros(64 * 4 * 8, dtype=torch.int64, device='cuda')
os.environ['TQ_FFN_PROF_PTR'] = hex(prof.data_ptr())
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)       # last launches: the 4-stage chain
torch.cuda.synchronize()
t = prof.cpu().reshape(64, 4, 8).double()
names = ['constants -> LDS', 'wait W1 + barrier A', 'GEMM 1 (+ issue W1 next)', 'epilogue 1', 'wait W2 + barrier B', 'GEMM 2 (+ issue W2 next)', 'tail']
for f in range(4):
    d = t[:, f, 1:] - t[:, f, :-1]
    print(f'stage {f}: ' + ', '.join(f'{n} {d[:, i].median():.0f}' for i, n in enumerate(names)) + f' | stage total {(t[:, f, 7] - t[:, f, 0]).median():.0f}')
print('first stamp -> last stamp of a block (median)', (t[:, 3, 7] - t[:, 0, 0]).median().item(), 'ticks (s_memtime: 100 MHz)')
