"""ctypes front end of oracle/tq_int_oracle.c: the INTEGER evaluation of fixed-range quantized layers on the CPU.

TEST INFRASTRUCTURE ONLY (tests/, tests/_oracle_backend.py): the checker of tq_linear_i8_fwd, tq_linear_i8_nonorm_fwd,
tq_ffn_i8_nonorm_fwd and tq_attention_i8_fwd.  Everything is exact integer contraction + single IEEE fp32 operations, so
the kernels are compared with it bit for bit (GELU excepted, see the C file's header).  Arguments mirror the backend
methods of quantization/_hip.py; tensors are CPU torch tensors, quantizers the 7-tuples
(delta, zero_float, signed, n_bits, symmetric, log_domain, eps) the backend takes."""
import ctypes as C

import numpy as np
import torch

from oracle import c_oracle


class _Q(C.Structure):
    _fields_ = [('delta', C.c_float), ('zero_float', C.c_float), ('eps', C.c_float), ('n_bits', C.c_int32),
                ('symmetric', C.c_int32), ('is_signed', C.c_int32), ('present', C.c_int32)]


_lib = None
_i8p, _fp, _i64 = C.POINTER(C.c_int8), C.POINTER(C.c_float), C.c_int64
_qp = C.POINTER(_Q)


def lib():
    global _lib
    if _lib is None:
        c_oracle.build()                       # make: no-op when up to date
        L = C.CDLL(c_oracle.LIB)
        L.tq_exp_neg.argtypes, L.tq_exp_neg.restype = [C.c_float], C.c_float
        L.tq_io_linear_i8.restype = None
        L.tq_io_linear_i8.argtypes = [_i8p, _i8p, _fp, _fp, _i8p, _i64, _i64, _i64, C.c_float, C.c_float, C.c_int,
                                      C.c_float, _fp, _i64, C.c_float, C.c_int, _qp, C.c_int, _fp, _fp, _fp, _qp, _qp]
        L.tq_io_ffn_i8.restype = None
        L.tq_io_ffn_i8.argtypes = [_i8p, C.c_float, C.c_float, C.c_int, C.c_float, _i8p, _fp, _fp, _i64, C.c_float, _qp,
                                   _i8p, _fp, _fp, _i64, C.c_float, _fp, _fp, _fp, _qp, _qp, _qp, _fp, _i8p, _i64, _i64,
                                   _i64, _i64]
        L.tq_io_attention_i8.restype = None
        L.tq_io_attention_i8.argtypes = [_i8p, _i8p, _i8p, _fp, _i8p, _i64, _i64, _i64, _i64, _i64, _fp, C.c_float,
                                         _qp, _qp, _qp, _qp, _qp, _qp]
        _lib = L
    return _lib


def _q(q):
    """backend quantizer tuple -> C struct (None -> absent)"""
    if q is None:
        return None, None
    delta, zf, signed, n_bits, symmetric, log_domain, eps = q
    assert not log_domain, 'integer oracle: linear scale domain only'
    s = _Q(float(delta), 0.0 if zf is None else float(zf), float(eps), int(n_bits), int(bool(symmetric)),
           int(bool(signed)) if signed is not None else 0, 1)
    return s, C.byref(s)


def _i8(t):
    a = np.ascontiguousarray(t.detach().cpu().numpy().astype(np.int8, copy=False))
    return a, a.ctypes.data_as(_i8p)


def _f(t):
    if t is None:
        return None, None
    a = np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32, copy=False))
    return a, a.ctypes.data_as(_fp)


def exp_neg(x):
    L = lib()
    return np.array([L.tq_exp_neg(float(v)) for v in np.asarray(x, np.float32).reshape(-1)], np.float32)


def linear_i8(x_idx, w_idx, bias, x_q, w_delta, w_eps, activation, q_out, tail=0, residual=None, nn_w=None, nn_b=None,
              q_t1=None, q_t2=None):
    """-> (y fp32 [M, N], y_idx int8 [M, N]); x_q = (delta, zero_float, n_bits, eps)"""
    L = lib()
    K = x_idx.shape[-1]
    M, N = x_idx.numel() // K, w_idx.shape[0]
    xa, xp = _i8(x_idx.reshape(M, K))
    wa, wp = _i8(w_idx)
    ba, bp = _f(bias)
    wd = w_delta.reshape(-1)
    da, dp = _f(wd)
    ra, rp = _f(None if residual is None else residual.reshape(M, N))
    na, np_ = _f(nn_w)
    nb, nbp = _f(nn_b)
    y = np.empty((M, N), np.float32)
    yi = np.empty((M, N), np.int8)
    so, po = _q(q_out)
    s1, p1 = _q(q_t1)
    s2, p2 = _q(q_t2)
    L.tq_io_linear_i8(xp, wp, bp, y.ctypes.data_as(_fp), yi.ctypes.data_as(_i8p), M, N, K, float(x_q[0]), float(x_q[1]),
                      int(x_q[2]), float(x_q[3]), dp, wd.numel(), float(w_eps), int(activation), po, int(tail), rp, np_,
                      nbp, p1, p2)
    shape = tuple(x_idx.shape[:-1]) + (N,)
    return torch.from_numpy(y).reshape(shape), torch.from_numpy(yi).reshape(shape)


def ffn_i8(x_idx, x_q, w1_idx, bias1, w1_delta, w1_eps, q_mid, w2_idx, bias2, w2_delta, w2_eps, residual, nn_w, nn_b,
           q_dense, q_sum, q_out):
    L = lib()
    K1 = x_idx.shape[-1]
    M, N1, N2 = x_idx.numel() // K1, w1_idx.shape[0], w2_idx.shape[0]
    xa, xp = _i8(x_idx.reshape(M, K1))
    w1a, w1p = _i8(w1_idx)
    w2a, w2p = _i8(w2_idx)
    b1a, b1p = _f(bias1)
    b2a, b2p = _f(bias2)
    d1 = w1_delta.reshape(-1)
    d2 = w2_delta.reshape(-1)
    d1a, d1p = _f(d1)
    d2a, d2p = _f(d2)
    ra, rp = _f(residual.reshape(M, N2))
    na, np_ = _f(nn_w)
    nb, nbp = _f(nn_b)
    sm, pm = _q(q_mid)
    sd, pd = _q(q_dense)
    ss, ps = _q(q_sum)
    so, po = _q(q_out)
    y = np.empty((M, N2), np.float32)
    yi = np.empty((M, N2), np.int8)
    L.tq_io_ffn_i8(xp, float(x_q[0]), float(x_q[1]), int(x_q[2]), float(x_q[3]), w1p, b1p, d1p, d1.numel(), float(w1_eps), pm,
                   w2p, b2p, d2p, d2.numel(), float(w2_eps), rp, np_, nbp, pd, ps, po, y.ctypes.data_as(_fp),
                   yi.ctypes.data_as(_i8p), M, K1, N1, N2)
    shape = tuple(x_idx.shape[:-1]) + (N2,)
    return torch.from_numpy(y).reshape(shape), torch.from_numpy(yi).reshape(shape)


def attention_i8(q_idx, k_idx, v_idx, num_heads, mask, denom, q_q, q_k, q_v, q_scores, q_probs, q_ctx, stride=None):
    """q/k/v int8(index - 128) [B, T, H * D] (or views into a stacked buffer with token stride `stride`)
    -> (ctx fp32 [B, T, H * D], ctx_idx int8)"""
    L = lib()
    B, T, HD = q_idx.shape
    D = HD // num_heads
    if stride is None:
        stride = HD
        qa, qp_ = _i8(q_idx)
        ka, kp = _i8(k_idx)
        va, vp = _i8(v_idx)
    else:
        raise NotImplementedError('pass contiguous q / k / v')
    ma, mp = _f(mask)
    ctx = np.empty((B, T, HD), np.float32)
    ci = np.empty((B, T, HD), np.int8)
    qs = [_q(t) for t in (q_q, q_k, q_v, q_scores, q_probs, q_ctx)]
    L.tq_io_attention_i8(qp_, kp, vp, ctx.ctypes.data_as(_fp), ci.ctypes.data_as(_i8p), B, T, num_heads, D, stride, mp,
                         float(denom), *[p for _, p in qs])
    return torch.from_numpy(ctx), torch.from_numpy(ci)
