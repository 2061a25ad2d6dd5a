#!/usr/bin/env python3
"""Build libtq_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is plain C ABI.

    python transformer-quantization_amd/build.py [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libtq_hip.so')
OBJDIR = os.path.join(HERE, 'build')

FLAGS = [
    '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
    '-ffp-contract=off',            # bit-exact parity: never fuse mul+add into fma
    '-fno-fast-math',
    '-fhip-fp32-correctly-rounded-divide-sqrt',   # IEEE division (x / scale)
    '-fno-gpu-rdc', '-Wall', '-Wno-unused-function',
]

# Per-source additions.  tq_attention_i8.hip: MFMA results in VGPRs.  The register allocator otherwise parks the i32
# accumulators of the attention core in AGPRs (36 per lane) and moves every one of them to a VGPR with its own
# v_accvgpr_read before the VALU epilogue can touch it (~100 moves per wave) -- and VGPRs + AGPRs together cost an
# occupancy step (T = 128: 147 -> 113 registers, 3 -> 4 waves per SIMD; key-split kernel 114 -> 78, 4 -> 5).
PER_SOURCE_FLAGS = {
    'tq_attention_i8.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form'],
}


def flags_for(src):
    return FLAGS + PER_SOURCE_FLAGS.get(os.path.basename(src), [])


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs.append(os.path.join(os.path.dirname(HERE), 'include', 'tq_hip.h'))
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    extra = os.environ.get('TQ_EXTRA_HIPCC_FLAGS', '').split()      # e.g. -DTQ_I8_DBG_BUILD for tools/tuning/i8_dbg.py
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = _headers()
    objs, procs = [], []
    for src in _sources():
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + flags_for(src) + extra + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {src}')
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


def build_fastcall(force=False, verbose=True):
    """The CPython stub over the C ABI (csrc_py/tq_fastcall.c: plain C, Python.h only) -> lib/_tq_fastcall<EXT_SUFFIX>.
    Optional at run time: quantization/_hip.py falls back to ctypes when it is absent."""
    import sysconfig
    src = os.path.join(HERE, 'csrc_py', 'tq_fastcall.c')
    out = os.path.join(LIBDIR, '_tq_fastcall' + (sysconfig.get_config_var('EXT_SUFFIX') or '.so'))
    os.makedirs(LIBDIR, exist_ok=True)
    if force or _stale(out, [src]):
        cmd = [os.environ.get('CC', 'gcc'), '-O2', '-fPIC', '-shared', '-Wall', '-I' + sysconfig.get_paths()['include'], src, '-o', out]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
    print(build_fastcall(force='--force' in sys.argv))
