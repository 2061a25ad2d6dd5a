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


RESOURCES = os.path.join(LIBDIR, 'kernel_resources.json')


def _parse_resources(stderr_text):
    """clang's -Rpass-analysis=kernel-resource-usage remarks -> {mangled kernel name: {vgpr, agpr, scratch, occupancy, lds}};
    everything else the compiler wrote to stderr (warnings, errors) is returned for printing."""
    import re
    res, cur, rest = {}, None, []
    for line in stderr_text.splitlines():
        m = re.search(r'remark:\s+(.*?)\s+\[-Rpass-analysis=kernel-resource-usage\]', line)
        if not m:
            if 'Rpass-analysis' not in line:
                rest.append(line)
            continue
        t = m.group(1)
        if t.startswith('Function Name:'):
            cur = res.setdefault(t.split(':', 1)[1].strip(), {})
        elif cur is not None and ':' in t:
            k, v = (x.strip() for x in t.split(':', 1))
            key = {'VGPRs': 'vgpr', 'AGPRs': 'agpr', 'ScratchSize [bytes/lane]': 'scratch', 'Occupancy [waves/SIMD]': 'occupancy',
                   'LDS Size [bytes/block]': 'lds'}.get(k)
            if key and v.isdigit():
                cur[key] = int(v)
    return res, '\n'.join(rest)


def build(force=False, verbose=True):
    """Compile stale sources, link lib/libtq_hip.so, and write lib/kernel_resources.json: registers, scratch bytes per lane,
    occupancy and LDS of EVERY kernel as the compiler reports them (tests/test_abi.py asserts that no kernel uses scratch
    memory: round 6 shipped, for two collections, a header change that made the register allocator spill the accumulators
    of the 128 x 128 tile Linear -- 2.5x slower at M = 8192 -- and nothing but the kernel table showed it)."""
    import json
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    extra = os.environ.get('TQ_EXTRA_HIPCC_FLAGS', '').split()      # e.g. -DTQ_I8_DBG_BUILD for tools/tuning/i8_dbg.py
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = _headers()
    objs, procs = [], []
    for src in _sources():
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs) or not os.path.exists(obj[:-2] + '.res.json'):
            cmd = [hipcc] + flags_for(src) + extra + ['-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, obj, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    for src, obj, p in procs:
        _, err = p.communicate()
        res, rest = _parse_resources(err or '')
        if rest.strip():
            print(rest, file=sys.stderr, flush=True)
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}')
        with open(obj[:-2] + '.res.json', 'w') as f:
            json.dump(res, f)
    if force or procs or _stale(LIB, objs) or not os.path.exists(RESOURCES):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        merged = {}
        for obj in objs:
            with open(obj[:-2] + '.res.json') as f:
                for k, v in json.load(f).items():
                    merged[k] = dict(v, source=os.path.basename(obj)[:-2] + '.hip')
        with open(RESOURCES, 'w') as f:
            json.dump(merged, f, indent=0, sort_keys=True)
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
