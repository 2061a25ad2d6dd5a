"""Error behaviour of the C ABI entry points of the fused paths: bad arguments come back as TQ_EINVAL /
TQ_EUNSUPPORTED with a message in tq_last_error() (no exceptions, no launch), zero-size problems are no-ops."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    from quantization import _hip
    be = _hip.backend()
    d = torch.tensor(0.1, device='cuda')
    z = torch.tensor(3.0, device='cuda')
    q = _hip.tq_quantizer(d.data_ptr(), z.data_ptr(), None, 8, 0, 0, 1e-8, 1, 1)
    return _hip, be, be.lib, q, (d, z)


def _err(lib):
    return lib.tq_last_error().decode()


def test_attention_entry_point(env):
    _hip, be, lib, q, keep = env
    x = torch.zeros(2, 64, 128, dtype=torch.int8, device='cuda')
    ctx = torch.empty(2, 64, 128, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    call = lambda **kw: lib.tq_attention_i8_fwd(
        kw.get('q', x.data_ptr()), x.data_ptr(), x.data_ptr(), ctx.data_ptr(), None, kw.get('B', 2), kw.get('T', 64), 2,
        kw.get('dh', 64), kw.get('stride', 0), None, kw.get('denom', 8.0), C.byref(q), C.byref(q), C.byref(q), None,
        kw.get('qp', C.byref(q)), None, st)
    assert call() == 0
    assert call(B=0) == 0                                               # empty batch: no-op
    assert call(q=None) == -1 and 'NULL' in _err(lib)
    assert call(T=96) == -1 and 'sequence length' in _err(lib)
    assert call(dh=48) == -1 and 'head_dim' in _err(lib)
    assert call(denom=0.0) == -1 and 'denom' in _err(lib)
    assert call(stride=100) == -1 and 'qkv_row_stride' in _err(lib)
    assert call(qp=None) == -1 and 'probabilities' in _err(lib)
    assert call(q=x.data_ptr() + 1) == -1 and 'alignment' in _err(lib)


def test_linear_entry_points(env):
    _hip, be, lib, q, (d, z) = env
    M, N, K = 64, 128, 128
    x = torch.zeros(M, K, dtype=torch.int8, device='cuda')
    w = torch.zeros(N, K, dtype=torch.int8, device='cuda')
    rs = torch.zeros(N, dtype=torch.int32, device='cuda')
    wd = torch.full((N,), 0.01, device='cuda')
    y = torch.empty(M, N, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    plain = lambda **kw: lib.tq_linear_i8_fwd(x.data_ptr(), w.data_ptr(), rs.data_ptr(), None, y.data_ptr(), None, 0,
                                              kw.get('M', M), kw.get('N', N), kw.get('K', K), d.data_ptr(), z.data_ptr(),
                                              kw.get('bits', 8), 1e-8, wd.data_ptr(), kw.get('wn', N), 1e-8, kw.get('act', 0),
                                              None, st)
    assert plain() == 0 and plain(M=0) == 0
    assert plain(K=100) == -1 and 'unsupported shape' in _err(lib)
    assert plain(bits=9) == -1 and '8 bits' in _err(lib)
    assert plain(wn=7) == -1 and 'weight scales' in _err(lib)
    assert plain(act=9) == -1 and 'activation' in _err(lib)
    arr = (C.POINTER(_hip.tq_quantizer) * 2)(C.pointer(q), C.pointer(q))
    grouped = lambda **kw: lib.tq_linear_i8_grouped_fwd(
        x.data_ptr(), w.data_ptr(), rs.data_ptr(), None, kw.get('y', y.data_ptr()), kw.get('yi', None), 0, M, N, K, d.data_ptr(),
        z.data_ptr(), 8, 1e-8, wd.data_ptr(), 1e-8, 0, kw.get('g', 2), C.cast(arr, C.POINTER(C.POINTER(_hip.tq_quantizer))), st)
    assert grouped() == 0
    assert grouped(g=4) == -1 and 'groups' in _err(lib)
    assert grouped(y=None) == -1 and 'no output' in _err(lib)


def test_tail_and_calibration_entry_points(env):
    _hip, be, lib, q, keep = env
    a = torch.zeros(8, 768, device='cuda')
    w = torch.ones(768, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    tail = lambda **kw: lib.tq_residual_nonorm_quant_fwd(a.data_ptr(), a.data_ptr(), a.data_ptr(), None, kw.get('rows', 8),
                                                         kw.get('d', 768), 0, C.byref(q), None, kw.get('w', w.data_ptr()),
                                                         w.data_ptr(), None, st)
    assert tail() == 0 and tail(rows=0) == 0
    assert tail(w=None) == -1 and 'NULL' in _err(lib)
    assert tail(d=772) == -4 and 'row length' in _err(lib)               # TQ_EUNSUPPORTED
    out = torch.empty(4, device='cuda')
    cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
    ws = torch.empty(1 << 16, dtype=torch.uint8, device='cuda')
    cal = lambda **kw: lib.tq_calibrate_tensor(a.data_ptr(), kw.get('n', a.numel()), 0, kw.get('mode', 0), None, None,
                                               out.data_ptr(), out.data_ptr() + 4, 0.9, 8, 0, 1e-8, 0, out.data_ptr() + 8,
                                               out.data_ptr() + 12, None, None, ws.data_ptr(), kw.get('wsb', ws.numel()),
                                               kw.get('cnt', cnt.data_ptr()), st)
    assert cal() == 0
    torch.cuda.synchronize()
    assert int(cnt) == 0, 'the ticket word must be back to 0 after the kernel'
    assert cal(n=0) == -1 and 'empty' in _err(lib)
    assert cal(mode=7) == -1 and 'mode' in _err(lib)
    assert cal(cnt=None) == -1
    assert cal(wsb=4) == -1 and 'workspace' in _err(lib)


def test_staircase_entry_points(env):
    _hip, be, lib, q, (d, z) = env
    st = torch.cuda.current_stream().cuda_stream
    nb = 768
    assert lib.tq_act_stair_bytes(nb) == 16 + 8 * nb
    table = torch.empty(lib.tq_act_stair_bytes(nb), dtype=torch.uint8, device='cuda')
    build = lambda **kw: lib.tq_act_stair_build(kw.get('act', 2), kw.get('q', C.byref(q)), kw.get('nb', nb),
                                                kw.get('table', table.data_ptr()), kw.get('bytes', table.numel()), st)
    assert build() == 0
    assert build(act=3) == -1 and 'no staircase' in _err(lib)                       # tanh: arithmetic path only
    assert build(nb=32) == -1 and 'bins' in _err(lib)
    assert build(nb=4096) == -1 and 'bins' in _err(lib)
    assert build(bytes=100) == -1 and 'too small' in _err(lib)
    assert build(table=None) == -1 and 'NULL' in _err(lib)
    assert build(table=table.data_ptr() + 4) == -1 and 'alignment' in _err(lib)
    q16 = _hip.tq_quantizer(d.data_ptr(), z.data_ptr(), None, 16, 0, 0, 1e-8, 1, 1)
    assert build(q=C.byref(q16)) == -1 and '8-bit' in _err(lib)
    # consumer: a table without an output quantizer, or with an impossible bin count, is refused; a table too large for
    # the kernel's LDS budget is simply not used (same result as without it)
    M, N, K = 64, 128, 128
    x = torch.zeros(M, K, dtype=torch.int8, device='cuda')
    w = torch.zeros(N, K, dtype=torch.int8, device='cuda')
    rs = torch.zeros(N, dtype=torch.int32, device='cuda')
    wd = torch.full((N,), 0.01, device='cuda')
    y = torch.empty(M, N, device='cuda')
    call = lambda **kw: lib.tq_linear_i8_stair_fwd(x.data_ptr(), w.data_ptr(), rs.data_ptr(), None, y.data_ptr(), None, 0, M, N, K,
                                                   d.data_ptr(), z.data_ptr(), 8, 1e-8, wd.data_ptr(), N, 1e-8, 2,
                                                   kw.get('q', C.byref(q)), kw.get('table', table.data_ptr()), kw.get('nb', nb), st)
    assert call() == 0
    assert call(q=None) == -1 and 'staircase' in _err(lib)
    assert call(nb=7) == -1 and 'staircase' in _err(lib)
    assert call(table=None, nb=0) == 0
    big = torch.empty(lib.tq_act_stair_bytes(2048), dtype=torch.uint8, device='cuda')
    assert lib.tq_act_stair_build(2, C.byref(q), 2048, big.data_ptr(), big.numel(), st) == 0
    y_ref = y.clone()
    assert call(table=big.data_ptr(), nb=2048) == 0 and torch.equal(y, y_ref)


def test_round5_entry_points(env):
    """tq_attention_i8_strided_fwd, tq_linear_i8_nonorm_grouped_fwd, tq_ffn_chain_i8_nonorm_fwd: argument errors."""
    _hip, be, lib, q, (d, z) = env
    st = torch.cuda.current_stream().cuda_stream
    x = torch.zeros(2, 64, 128, dtype=torch.int8, device='cuda')
    ctx = torch.empty(2, 64, 128, device='cuda')
    att = lambda vs: lib.tq_attention_i8_strided_fwd(x.data_ptr(), x.data_ptr(), x.data_ptr(), ctx.data_ptr(), None, 2, 64, 2, 64, 0,
                                                     vs, None, 8.0, C.byref(q), C.byref(q), C.byref(q), None, C.byref(q), None, st)
    assert att(0) == 0 and att(128) == 0
    assert att(100) == -1 and 'v_row_stride' in _err(lib)
    # grouped Linear -> NoNorm chains
    M, N, K = 64, 128, 128
    xi = torch.zeros(M, K, dtype=torch.int8, device='cuda')
    w = torch.zeros(2 * N, K, dtype=torch.int8, device='cuda')
    rs = torch.zeros(2 * N, dtype=torch.int32, device='cuda')
    wd = torch.full((2 * N,), 0.01, device='cuda')
    nw, nb = torch.ones(2 * N, device='cuda'), torch.zeros(2 * N, device='cuda')
    y = torch.empty(2, M, N, device='cuda')
    QP = C.POINTER(_hip.tq_quantizer)
    two = C.cast((QP * 2)(C.pointer(q), C.pointer(q)), C.POINTER(QP))
    one = C.cast((QP * 2)(C.pointer(q), None), C.POINTER(QP))
    grp = lambda **kw: lib.tq_linear_i8_nonorm_grouped_fwd(xi.data_ptr(), w.data_ptr(), rs.data_ptr(), None, nw.data_ptr(), nb.data_ptr(),
                                                         y.data_ptr(), None, 0, M, kw.get('N', 2 * N), K, d.data_ptr(), z.data_ptr(), 8,
                                                         1e-8, wd.data_ptr(), 1e-8, kw.get('G', 2), kw.get('qd', two), kw.get('qo', two), st)
    assert grp() == 0
    assert grp(G=4) == -1 and '2 or 3 groups' in _err(lib)
    assert grp(N=2 * N - 64) == -1
    assert grp(qd=one) == -1 and 'dense-output quantizer' in _err(lib)
    assert grp(qo=one) == -1 and 'output quantizer' in _err(lib)
    # chain of feed-forward blocks
    w1 = torch.zeros(512, 128, dtype=torch.int8, device='cuda'); w2 = torch.zeros(128, 512, dtype=torch.int8, device='cuda')
    rs1 = torch.zeros(512, dtype=torch.int32, device='cuda'); rs2 = torch.zeros(128, dtype=torch.int32, device='cuda')
    s1 = torch.full((1,), 0.01, device='cuda')
    n1, n0 = torch.ones(128, device='cuda'), torch.zeros(128, device='cuda')
    xf = torch.zeros(32, 128, dtype=torch.int8, device='cuda'); res = torch.zeros(32, 128, device='cuda'); yo = torch.empty(32, 128, device='cuda')

    def stage(**kw):
        g = _hip.tq_ffn_stage()
        g.w1_idx, g.w1_rowsum, g.w1_delta, g.w1_n_params, g.w1_eps = kw.get('w1', w1.data_ptr()), rs1.data_ptr(), s1.data_ptr(), 1, 1e-8
        g.q_mid = C.pointer(q)
        g.w2_idx, g.w2_rowsum, g.w2_delta, g.w2_n_params, g.w2_eps = w2.data_ptr(), rs2.data_ptr(), s1.data_ptr(), 1, 1e-8
        g.nn_weight, g.nn_bias = n1.data_ptr(), n0.data_ptr()
        g.q_out = kw.get('q_out', C.pointer(q))
        return g
    chain = lambda stages, n=None: lib.tq_ffn_chain_i8_nonorm_fwd(
        xf.data_ptr(), d.data_ptr(), z.data_ptr(), 8, 1e-8, res.data_ptr(), C.cast((_hip.tq_ffn_stage * len(stages))(*stages), C.c_void_p),
        len(stages) if n is None else n, yo.data_ptr(), None, 0, 32, 128, 512, 128, st)
    assert chain([stage(), stage()]) == 0
    assert chain([stage()]) == -1 and '2..4' in _err(lib)
    assert chain([stage()] * 4, n=5) == -1 and '2..4' in _err(lib)
    assert chain([stage(w1=None), stage()]) == -1 and 'NULL pointer in stage 0' in _err(lib)
    assert chain([stage(q_out=None), stage()]) == -1 and 'stage 0 needs an asymmetric' in _err(lib)
    assert chain([stage(), stage(q_out=None)]) == 0                      # the last block may go without output quantizer
    torch.cuda.synchronize()
