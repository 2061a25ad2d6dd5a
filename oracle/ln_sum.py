"""TEST INFRASTRUCTURE.  LayerNorm whose fp32 statistics are summed in the order of the fused tail kernel
(`res_ln_quant_k`, csrc/tq_fused_ln.hip), restated in plain C in oracle/tq_ln_oracle.c -- the counterpart of
oracle/aten_sum.py (which pins ATen's cascade sum for the MSE losses): with it the f2 chain
Q(LayerNorm(Q(Q(a) + r))) (reference models/quantized_bert.py:238-248, 264-280) is comparable with the kernel at ZERO
tolerance (tests/test_fused_ln.py).  `torch.nn.functional.layer_norm` -- what the reference runs -- leaves that order to
the backend; against IT the kernel keeps its tolerance contract (same file).
"""
import ctypes as C

import numpy as np
import torch

from oracle import c_oracle

# launch_res_ln's table (csrc/tq_fused_ln.hip): vectors per row -> (lanes per row, vectors per lane), first match
_TABLE = ((32, 3), (64, 3), (64, 6), (64, 12), (64, 1), (64, 2), (64, 4), (16, 1), (32, 1), (64, 8))


def kernel_layout(d, storage_dtype):
    V = 4 if storage_dtype == torch.float32 else 8
    assert d % V == 0
    vpr = d // V
    for lpr, nv in _TABLE:
        if vpr == lpr * nv:
            return lpr, nv, V
    raise ValueError(f'row length {d} has no instantiation')


def layer_norm_kernel_order(u, weight, bias, eps, storage_dtype):
    """u: fp32 [rows, d] (the quantized residual sum as the kernel holds it in registers) -> fp32 [rows, d]."""
    lib = c_oracle.load()
    fn = lib.tq_oracle_layernorm_kernel_order
    fp = C.POINTER(C.c_float)
    fn.argtypes = [fp, fp, C.c_int64, C.c_int, C.c_int, C.c_int, fp, fp, C.c_float]
    fn.restype = None
    rows, d = u.shape
    lpr, nv, V = kernel_layout(d, storage_dtype)
    x = np.ascontiguousarray(u.detach().cpu().numpy(), dtype=np.float32)
    w = np.ascontiguousarray(weight.detach().cpu().numpy(), dtype=np.float32)
    b = np.ascontiguousarray(bias.detach().cpu().numpy(), dtype=np.float32)
    out = np.empty_like(x)
    fn(x.ctypes.data_as(fp), out.ctypes.data_as(fp), rows, lpr, nv, V, w.ctypes.data_as(fp), b.ctypes.data_as(fp),
       float(np.float32(eps)))
    return torch.from_numpy(out)
