"""ctypes loader for the plain-C oracle (oracle/tq_oracle_core.c).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, '_build', 'libtq_oracle.so')


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])
    return LIB


def load():
    if not os.path.exists(LIB):
        build()
    lib = C.CDLL(LIB)
    fp = C.POINTER(C.c_float)
    lib.tq_oracle_fake_quant_f32.argtypes = [fp, fp, fp, C.c_size_t, fp, fp, C.c_int, C.c_int,
                                             C.c_float, C.c_size_t, C.c_size_t]
    lib.tq_oracle_fake_quant_f32.restype = None
    lib.tq_oracle_minmax_f32.argtypes = [fp, C.c_size_t, fp, fp]
    lib.tq_oracle_range_to_asym.argtypes = [C.c_float, C.c_float, C.c_int, C.c_float, fp, fp]
    lib.tq_oracle_range_to_sym.argtypes = [C.c_float, C.c_float, C.c_int, C.c_float, fp,
                                           C.POINTER(C.c_int)]
    lib.tq_oracle_mse_f32.argtypes = [fp, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float]
    lib.tq_oracle_mse_f32.restype = C.c_double
    return lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def fake_quant(x, delta, zero_float, signed, n_bits, eps=1e-8, n_params=1, inner=1):
    """numpy fp32 in -> (idx, y) numpy fp32."""
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    delta = np.ascontiguousarray(np.atleast_1d(delta), dtype=np.float32)
    zf = None if zero_float is None else np.ascontiguousarray(np.atleast_1d(zero_float),
                                                              dtype=np.float32)
    y, idx = np.empty_like(x), np.empty_like(x)
    lib.tq_oracle_fake_quant_f32(_fp(x), _fp(y), _fp(idx), x.size, _fp(delta),
                                 None if zf is None else _fp(zf), int(bool(signed)), n_bits, eps,
                                 n_params, inner)
    return idx, y
