"""Build a second copy of the library from the CURRENT sources with ONE file replaced (A/B of two builds on one box through
TQ_LIB_PATH):   python tools/tuning/build_ab_lib.py csrc/tq_device.h /tmp/tq_device_r5.h  ->  tools/tuning/_ab/libtq_hip.so"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, 'transformer-quantization_amd')
sys.path.insert(0, PKG)
import build as B
rel, repl = sys.argv[1], sys.argv[2]
OUT = os.path.join(ROOT, 'tools', 'tuning', '_ab')
SRC = os.path.join(OUT, 'pkg', 'csrc')
shutil.rmtree(OUT, ignore_errors=True)
shutil.copytree(os.path.join(PKG, 'csrc'), SRC)
shutil.copy(repl, os.path.join(OUT, 'pkg', rel))
shutil.copytree(os.path.join(ROOT, 'include'), os.path.join(OUT, 'include'))
srcs = sorted(os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith('.hip'))
procs = [subprocess.Popen(['/opt/rocm/bin/hipcc'] + B.flags_for(s) + ['-c', s, '-o', s[:-4] + '.o']) for s in srcs]
assert all(p.wait() == 0 for p in procs)
lib = os.path.join(OUT, 'libtq_hip.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + [s[:-4] + '.o' for s in srcs])
for f in os.listdir(os.path.join(PKG, 'lib')):
    if f.startswith('_tq_fastcall'):
        shutil.copy(os.path.join(PKG, 'lib', f), OUT)
print(lib)
