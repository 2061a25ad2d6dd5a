"""Why `scale_domain='log'` keeps a TOLERANCE contract (VERDICT r4 missing #3) -- the evidence, pinned.

The reference evaluates a log-domain quantizer's scale as `torch.exp(_delta)` with `_delta = torch.log(delta)`
(quantization/quantizers.py:143-147, 279-282).  On the torch build the reference fixtures were generated with (2.10.0,
USE_MKL=ON) ATen's CPU float32 exp / log are Intel MKL VML `vsExp` / `vsLn` (aten/src/ATen/cpu/vml.h, IMPLEMENT_VML_MKL) --
a closed-source third-party library: there is no published algorithm to restate on the device, and the bits differ from
both SLEEF's expf (what a torch build WITHOUT MKL would run) and glibc's expf in about 1-8 % of inputs.  The reference's
own log-domain scale is therefore build-dependent in its last bit; the HIP path uses the correctly rounded-to-1-ulp
device exp / log and the parity tests assert >= 99.9 % identical indices (tests/test_hip_parity.py).  This test pins the
provenance claim; it skips on a torch build that does not export MKL's entry points."""
import ctypes as C
import os

import numpy as np
import pytest
import torch


def _mkl():
    try:
        lib = C.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libtorch_cpu.so'))
        return lib.vsExp, lib.vsLn
    except (OSError, AttributeError):
        return None


def test_torch_cpu_exp_and_log_are_mkl_vml():
    fns = _mkl()
    if fns is None:
        pytest.skip('this torch build does not export MKL VML (vsExp / vsLn)')
    vs_exp, vs_ln = fns
    fp = C.POINTER(C.c_float)
    vs_exp.argtypes = vs_ln.argtypes = [C.c_int, fp, fp]
    rs = np.random.RandomState(1)
    x = rs.uniform(-20, 5, 100000).astype(np.float32)
    y = np.empty_like(x)
    vs_exp(x.size, x.ctypes.data_as(fp), y.ctypes.data_as(fp))
    assert np.array_equal(y, torch.exp(torch.from_numpy(x)).numpy())
    assert all(float(torch.exp(torch.tensor(v))) == w for v, w in zip(x[:200], y[:200]))      # 0-d tensors: the same path
    xp = np.exp(rs.uniform(-30, 20, 100000)).astype(np.float32)
    vs_ln(xp.size, xp.ctypes.data_as(fp), y.ctypes.data_as(fp))
    assert np.array_equal(y, torch.log(torch.from_numpy(xp)).numpy())
    # ... and it is NOT the libm every other consumer would use: glibc's expf differs in ~1 % of these inputs
    libm = C.CDLL('libm.so.6')
    libm.expf.restype, libm.expf.argtypes = C.c_float, [C.c_float]
    ref = torch.exp(torch.from_numpy(x[:20000])).numpy()
    differ = sum(np.float32(libm.expf(float(v))) != r for v, r in zip(x[:20000], ref))
    assert 0 < differ < 2000, differ
