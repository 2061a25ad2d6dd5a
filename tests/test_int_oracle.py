"""The integer evaluation of fixed-range quantized layers, pinned at ZERO tolerance: tq_linear_i8_fwd,
tq_linear_i8_nonorm_fwd, tq_ffn_i8_nonorm_fwd and tq_attention_i8_fwd against oracle/tq_int_oracle.c (exact integer
contraction + single IEEE fp32 operations; the softmax exponential is the IEEE-only `exp_neg_ieee` on both sides).

* CPU: the integer oracle itself is pinned against a float64 simulation of the reference's formula
  (hijacker.py:66-116: F.linear on the dequantised tensors, then the activation and the output quantizer): the two can
  differ only where the float64 value sits within fp32 round-off of a rounding tie.
* GPU: every output (fp32 values AND int8 indices) of the kernels equals the oracle's bit for bit, at the BERT-base and
  MobileBERT shapes -- since round 4 also with GELU, which in front of a <= 8-bit quantizer is the correctly rounded fp32
  GELU evaluated through a staircase table (csrc/tq_stair.hip; oracle activation code 4 evaluates the same
  specification directly).  Only the arithmetic GELU epilogue (hardware v_exp_f32 inside an erf fit; used where the table
  declines) is held to a tolerance: <= 1 grid step on <= 2e-5 of the outputs."""
import numpy as np
import pytest
import torch

from oracle import int_oracle as IO

DEV = 'cuda'


def _rand_layer(M, N, K, seed, x_bits=8, w_bits=8, o_bits=8, per_row=False):
    g = torch.Generator().manual_seed(seed)
    x_idx = torch.randint(0, 2 ** x_bits, (M, K), generator=g)
    w_idx = torch.randint(-(2 ** (w_bits - 1)) + 1, 2 ** (w_bits - 1), (N, K), generator=g)
    x_q = (0.02 + 0.001 * (seed % 7), float(2 ** (x_bits - 1) - 11), x_bits, 1e-8)        # delta, zero_float, n_bits, eps
    w_delta = (torch.rand(N if per_row else 1, generator=g) * 0.002 + 0.0005)
    bias = torch.randn(N, generator=g) * 0.1
    # output grid sized to the pre-activation spread
    spread = float(np.sqrt(K) * 2 ** x_bits / 3.5 * 2 ** (w_bits - 1) / 1.7 * x_q[0] * float(w_delta.mean()))
    q_out = (torch.tensor(2 * spread / (2 ** o_bits - 1)), torch.tensor(float(2 ** (o_bits - 1))), None, o_bits, False, False,
             1e-8)
    return (x_idx - 128).to(torch.int8), w_idx.to(torch.int8), x_q, w_delta, bias, q_out


def _f64_sim(x_idx, w_idx, x_q, w_delta, bias, q_out, act):
    """float64 simulation of the reference formula on the dequantised tensors"""
    zp = float(np.clip(np.rint(x_q[1]), 0, 2 ** x_q[2] - 1))
    xd = (x_idx.double() + 128 - zp) * float(np.float32(x_q[0]))
    wd = w_idx.double() * w_delta.double().reshape(-1, 1)
    v = xd @ wd.t() + bias.double()
    if act == 1:
        v = torch.relu(v)
    if act in (2, 4):                                   # nn.GELU(): the erf form, here in float64
        v = torch.nn.functional.gelu(v)
    d, zf, _, nb, *_ = q_out
    s = float(d)
    z = float(np.clip(np.rint(float(zf)), 0, 2 ** nb - 1))
    idx = torch.clamp(torch.round(v / s) + z, 0, 2 ** nb - 1)
    return idx, s * (idx - z)


@pytest.mark.parametrize('act', [0, 1, 2, 4], ids=['none', 'relu', 'gelu-fit', 'gelu-exact'])
@pytest.mark.parametrize('shape', [(64, 128, 512), (32, 512, 128), (96, 96, 768)])
def test_integer_oracle_matches_float64_simulation(shape, act):
    M, N, K = shape
    x_idx, w_idx, x_q, w_delta, bias, q_out = _rand_layer(M, N, K, seed=11 + N + act, o_bits=8)
    y, yi = IO.linear_i8(x_idx, w_idx, bias, x_q, w_delta, 1e-8, act, tuple(
        float(v) if torch.is_tensor(v) else v for v in q_out))
    idx_ref, y_ref = _f64_sim(x_idx, w_idx, x_q, w_delta, bias, q_out, act)
    got = yi.double() + 128
    assert float((got != idx_ref).double().mean()) <= 2e-3           # fp32-vs-fp64 round-off at rounding ties only
    assert float((got - idx_ref).abs().max()) <= 1
    assert torch.allclose(y.double(), y_ref, atol=float(q_out[0]) * 1.001, rtol=0)
    # the contraction itself is exact: turning the quantizer off, the oracle's value is the fp32 rounding of the exact
    # rational s_x s_w (sum) + b up to the two fp32 roundings of the epilogue
    y0, _ = IO.linear_i8(x_idx, w_idx, bias, x_q, w_delta, 1e-8, 0, None)
    zp = float(np.clip(np.rint(x_q[1]), 0, 2 ** x_q[2] - 1))
    exact = ((x_idx.double() + 128 - zp) @ w_idx.double().t()) * float(np.float32(x_q[0])) * float(w_delta[0]) + bias.double()
    assert torch.allclose(y0.double(), exact, rtol=3e-7, atol=1e-7)


def test_exp_neg_is_accurate_and_total():
    x = np.concatenate([np.linspace(-86, 0, 20001), -np.logspace(-6, 1.9, 2000)]).astype(np.float32)
    got = IO.exp_neg(x).astype(np.float64)
    ref = np.exp(x.astype(np.float64))
    assert (np.abs(got - ref) / ref).max() <= 2 * 2.0 ** -24
    assert IO.exp_neg([0.0])[0] == 1.0 and IO.exp_neg([-87.0])[0] == 0.0 and IO.exp_neg([-np.inf])[0] == 0.0
    assert np.isnan(IO.exp_neg([np.nan])[0])


# ------------------------------------------------------------------------------------------------------------ GPU
def _dev(q):
    if q is None:
        return None
    return tuple(v.to(DEV) if torch.is_tensor(v) else v for v in q)


def _xq_dev(x_q):
    return (torch.tensor(x_q[0], device=DEV), torch.tensor(x_q[1], device=DEV), x_q[2], x_q[3])


LIN_SHAPES = [(1024, 768, 768), (1024, 3072, 768), (1024, 768, 3072),        # BERT-base
              (1024, 128, 512), (1024, 512, 128), (1024, 128, 128), (1024, 512, 512), (256, 512, 384)]   # MobileBERT


@pytest.mark.gpu
@pytest.mark.parametrize('o_bits', [8, 4])
@pytest.mark.parametrize('act', [0, 1], ids=['none', 'relu'])
@pytest.mark.parametrize('shape', LIN_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_linear_i8_equals_integer_oracle_bit_for_bit(shape, act, o_bits):
    from quantization import _hip
    be = _hip.backend()
    M, N, K = shape
    per_row = N == 768
    x_idx, w_idx, x_q, w_delta, bias, q_out = _rand_layer(M, N, K, seed=3 + N + K + act, o_bits=o_bits, per_row=per_row)
    ref_y, ref_i = IO.linear_i8(x_idx, w_idx, bias, x_q, w_delta, 1e-8, act, tuple(
        float(v) if torch.is_tensor(v) else v for v in q_out))
    wi = w_idx.to(DEV)
    y, yi = be.linear_i8(x_idx.to(DEV), wi, be.rowsum_i8(wi), bias.to(DEV), _xq_dev(x_q), w_delta.to(DEV), 1e-8, act,
                         _dev(q_out), torch.float32, want_idx=True)
    assert torch.equal(yi.cpu(), ref_i)
    assert torch.equal(y.cpu(), ref_y)
    # index-only output (y = NULL): the same indices
    _, yi2 = be.linear_i8(x_idx.to(DEV), wi, be.rowsum_i8(wi), bias.to(DEV), _xq_dev(x_q), w_delta.to(DEV), 1e-8, act,
                          _dev(q_out), torch.float32, want_idx=True, want_y=False)
    assert _ is None and torch.equal(yi2.cpu(), ref_i)
    # without an output quantizer: the raw fp32 pre-activation
    ref0, _ = IO.linear_i8(x_idx, w_idx, bias, x_q, w_delta, 1e-8, act, None)
    y0 = be.linear_i8(x_idx.to(DEV), wi, be.rowsum_i8(wi), bias.to(DEV), _xq_dev(x_q), w_delta.to(DEV), 1e-8, act, None,
                      torch.float32)
    assert torch.equal(y0.cpu(), ref0)


@pytest.mark.gpu
def test_linear_i8_gelu_vs_integer_oracle():
    """GELU: same erf fits, but exp2 is v_exp_f32 on the GPU and libm's exp2f in the oracle (<= 1 ulp apart): the
    8-bit indices agree except where that last bit decides a rounding tie."""
    from quantization import _hip
    be = _hip.backend()
    M, N, K = 1024, 3072, 768
    x_idx, w_idx, x_q, w_delta, bias, q_out = _rand_layer(M, N, K, seed=77)
    ref_y, ref_i = IO.linear_i8(x_idx, w_idx, bias, x_q, w_delta, 1e-8, 2, tuple(
        float(v) if torch.is_tensor(v) else v for v in q_out))
    wi = w_idx.to(DEV)
    y, yi = be.linear_i8(x_idx.to(DEV), wi, be.rowsum_i8(wi), bias.to(DEV), _xq_dev(x_q), w_delta.to(DEV), 1e-8, 2,
                         _dev(q_out), torch.float32, want_idx=True)
    d = (yi.cpu().int() - ref_i.int()).abs()
    assert int(d.max()) <= 1 and float((d != 0).float().mean()) <= 1e-5, (int(d.max()), float((d != 0).float().mean()))


def _f(q):
    return None if q is None else tuple(float(v) if torch.is_tensor(v) else v for v in q)


# ---- GELU + quantizer as a staircase table (csrc/tq_stair.hip): the specification is the CORRECTLY ROUNDED fp32 GELU in
# front of the reference quantizer (oracle act code 4), and the table reproduces it for every fp32 pre-activation
def _stair_header(stair):
    return stair[0][:16].view(torch.float32).cpu().tolist()          # 1 / bin width, offset, n_bins - 1, ok


def _gelu_q(scale, zero, bits=8):
    return (torch.tensor(float(scale)), torch.tensor(float(zero)), None, bits, False, False, 1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(1024, 3072, 768), (4096, 4096, 128)], ids=['64x64-tiles', '128x128-tiles'])
def test_linear_i8_gelu_staircase_equals_the_exact_gelu_oracle_bit_for_bit(shape):
    from quantization import _hip
    be = _hip.backend()
    M, N, K = shape
    x_idx, w_idx, x_q, w_delta, bias, q_sym = _rand_layer(M, N, K, seed=91 + K)
    q_out = _gelu_q(float(q_sym[0]) * 0.55, 9.0)                     # GELU outputs: [-0.17, ~spread]
    stair = be.act_stair(2, _dev(q_out))
    assert _stair_header(stair)[3] == 1.0, _stair_header(stair)
    ref_y, ref_i = IO.linear_i8(x_idx, w_idx, bias, x_q, w_delta, 1e-8, 4, _f(q_out))
    wi = w_idx.to(DEV)
    args = (x_idx.to(DEV), wi, be.rowsum_i8(wi), bias.to(DEV), _xq_dev(x_q), w_delta.to(DEV), 1e-8, 2, _dev(q_out), torch.float32)
    y, yi = be.linear_i8(*args, want_idx=True, stair=stair)
    assert torch.equal(yi.cpu(), ref_i) and torch.equal(y.cpu(), ref_y)
    assert ref_i.unique().numel() > 100                              # the case exercises most of the grid
    _, yi2 = be.linear_i8(*args, want_idx=True, want_y=False, stair=stair)
    assert torch.equal(yi2.cpu(), ref_i)
    # and the arithmetic epilogue (single erf fit, 8.5e-8 absolute) stays within its documented distance of the exact one
    _, ya = be.linear_i8(*args, want_idx=True)
    d = (ya.cpu().int() - ref_i.int()).abs()
    assert int(d.max()) <= 1 and float((d != 0).float().mean()) <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize('q_out', [_gelu_q(0.036, 5.0), _gelu_q(0.0131, 13.0), _gelu_q(0.25, 1.0, 4), _gelu_q(0.02, 200.0),
                                   (torch.tensor(0.05), None, torch.tensor(True), 8, True, False, 1e-8),
                                   (torch.tensor(0.03), None, torch.tensor(False), 8, True, False, 1e-8),
                                   _gelu_q(1.3, 0.0, 2)],
                         ids=['s0.036', 's0.0131', '4bit', 'mostly-clamped', 'symmetric-signed', 'symmetric-unsigned', '2bit'])
def test_gelu_staircase_is_exact_at_every_step_of_the_table(q_out):
    """Zero weights make the pre-activation of column n the bias b[n] exactly, so ANY fp32 value can be pushed through the
    table: every threshold the builder found, its two fp32 neighbours, the bin edges, the neighbourhood of GELU's
    minimum, zeros, denormals, huge values and a dense random sample -- all equal to the direct evaluation."""
    from quantization import _hip
    be = _hip.backend()
    stair = be.act_stair(2, _dev(q_out))
    inv_w, c0, nbm1, ok = _stair_header(stair)
    assert ok == 1.0
    tab = stair[0][16:].view(torch.int32).cpu().numpy().reshape(-1, 2)
    T = tab[:, 0].copy().view(np.float32)
    T = T[np.isfinite(T) & (np.abs(T) < 1e30)]
    edges = ((np.arange(int(nbm1) + 2) - np.float64(c0)) / np.float64(inv_w)).astype(np.float32)
    rng = np.random.RandomState(5)
    pts = [T, np.nextafter(T, np.float32(-np.inf)), np.nextafter(T, np.float32(np.inf)), edges,
           np.nextafter(edges, np.float32(-np.inf)), np.nextafter(edges, np.float32(np.inf)),
           np.float32(-0.7517916) + np.arange(-64, 65, dtype=np.float32) * np.float32(2.0 ** -23),
           np.array([0.0, -0.0, 1e-40, -1e-40, 1e30, -1e30, 3.4e38, -3.4e38, 1e-20, -1e-20], np.float32),
           (rng.randn(20000) * 3).astype(np.float32), (rng.rand(8000) * 14 - 8).astype(np.float32)]
    v = np.concatenate([np.asarray(a, np.float32).ravel() for a in pts])
    N = -(-v.size // 64) * 64
    bias = torch.from_numpy(np.concatenate([v, np.zeros(N - v.size, np.float32)]))
    M, K = 64, 128
    x_idx = torch.zeros(M, K, dtype=torch.int8)
    w_idx = torch.zeros(N, K, dtype=torch.int8)
    x_q, w_delta = (0.02, 117.0, 8, 1e-8), torch.tensor([0.001])
    ref_y, _ = IO.linear_i8(x_idx, w_idx, bias, x_q, w_delta, 1e-8, 4, _f(q_out))
    wi = w_idx.to(DEV)
    y = be.linear_i8(x_idx.to(DEV), wi, be.rowsum_i8(wi), bias.to(DEV), _xq_dev(x_q), w_delta.to(DEV), 1e-8, 2, _dev(q_out),
                     torch.float32, stair=stair)
    bad = (y.cpu() != ref_y).nonzero()
    assert bad.numel() == 0, [(float(bias[j]), float(y[i, j]), float(ref_y[i, j])) for i, j in bad[:5].tolist()]


@pytest.mark.gpu
@pytest.mark.parametrize('act', [0, 1], ids=['none', 'relu'])
def test_staircase_of_identity_and_relu_equals_the_arithmetic_epilogue(act):
    """tq_act_stair_build also tabulates the quantizer alone and ReLU + quantizer (the Python layer only uses it for
    GELU, where it pays).  There the correctly rounded activation IS the fp32 activation, so the table must reproduce the
    arithmetic epilogue -- and the oracle -- bit for bit, at every threshold and its neighbours."""
    from quantization import _hip
    be = _hip.backend()
    q_out = _gelu_q(0.021, 97.0)
    stair = be.act_stair(act, _dev(q_out))
    inv_w, c0, nbm1, ok = _stair_header(stair)
    assert ok == 1.0
    tab = stair[0][16:].view(torch.int32).cpu().numpy().reshape(-1, 2)
    T = tab[:, 0].copy().view(np.float32)
    T = T[np.isfinite(T) & (np.abs(T) < 1e30)]
    rng = np.random.RandomState(6)
    v = np.concatenate([T, np.nextafter(T, np.float32(-np.inf)), np.nextafter(T, np.float32(np.inf)),
                        (rng.randn(8000) * 2).astype(np.float32), np.array([0.0, -0.0, 1e30, -1e30, 1e-40], np.float32)])
    N = -(-v.size // 64) * 64
    bias = torch.from_numpy(np.concatenate([v, np.zeros(N - v.size, np.float32)]))
    M, K = 64, 128
    x_idx, w_idx = torch.zeros(M, K, dtype=torch.int8), torch.zeros(N, K, dtype=torch.int8)
    x_q, w_delta = (0.02, 117.0, 8, 1e-8), torch.tensor([0.001])
    ref_y, ref_i = IO.linear_i8(x_idx, w_idx, bias, x_q, w_delta, 1e-8, act, _f(q_out))
    wi = w_idx.to(DEV)
    args = (x_idx.to(DEV), wi, be.rowsum_i8(wi), bias.to(DEV), _xq_dev(x_q), w_delta.to(DEV), 1e-8, act, _dev(q_out), torch.float32)
    y, yi = be.linear_i8(*args, want_idx=True, stair=stair)
    y0, yi0 = be.linear_i8(*args, want_idx=True)
    assert torch.equal(yi.cpu(), ref_i) and torch.equal(y.cpu(), ref_y)
    assert torch.equal(yi, yi0) and torch.equal(y, y0) and ref_i.unique().numel() > 100


@pytest.mark.gpu
def test_gelu_staircase_declines_a_grid_it_cannot_hold():
    """A grid much finer than the bins: the builder says so in the header and the consumer keeps its arithmetic epilogue
    (identical output to the call without a table)."""
    from quantization import _hip
    be = _hip.backend()
    assert _stair_header(be.act_stair(2, _dev(_gelu_q(1e-9, 3.0))))[3] == 0.0      # delta below eps: scale = eps = 1e-8
    q_out = _gelu_q(0.0009, 190.0)
    stair = be.act_stair(2, _dev(q_out))
    assert _stair_header(stair)[3] == 0.0
    M, N, K = 256, 512, 256
    x_idx, w_idx, x_q, w_delta, bias, _ = _rand_layer(M, N, K, seed=12)
    w_delta = w_delta * 0.05
    wi = w_idx.to(DEV)
    args = (x_idx.to(DEV), wi, be.rowsum_i8(wi), bias.to(DEV), _xq_dev(x_q), w_delta.to(DEV), 1e-8, 2, _dev(q_out), torch.float32)
    y0, i0 = be.linear_i8(*args, want_idx=True)
    y1, i1 = be.linear_i8(*args, want_idx=True, stair=stair)
    assert torch.equal(y0, y1) and torch.equal(i0, i1) and i0.unique().numel() > 50


def _tail_args(N, seed):
    g = torch.Generator().manual_seed(seed)
    nn_w = torch.rand(N, generator=g) + 0.5
    nn_b = torch.randn(N, generator=g) * 0.1
    q_sum = (torch.tensor(0.07), torch.tensor(121.0), None, 8, False, False, 1e-8)
    q_fin = (torch.tensor(0.6), torch.tensor(7.0), None, 4, False, False, 1e-8)
    return nn_w, nn_b, q_sum, q_fin


@pytest.mark.gpu
@pytest.mark.parametrize('with_residual', [False, True], ids=['bottleneck', 'residual'])
@pytest.mark.parametrize('shape', [(1024, 128, 512), (1024, 512, 128), (1024, 128, 128)], ids=lambda s: 'x'.join(map(str, s)))
def test_linear_i8_nonorm_equals_integer_oracle(shape, with_residual):
    from quantization import _hip
    be = _hip.backend()
    M, N, K = shape
    x_idx, w_idx, x_q, w_delta, bias, q_dense = _rand_layer(M, N, K, seed=5 + N + K, o_bits=4)
    nn_w, nn_b, q_sum, q_fin = _tail_args(N, 9)
    res = torch.randn(M, N, generator=torch.Generator().manual_seed(4)) if with_residual else None
    ref_y, ref_i = IO.linear_i8(x_idx, w_idx, bias, x_q, w_delta, 1e-8, 0, _f(q_dense), tail=2 if with_residual else 1,
                                residual=res, nn_w=nn_w, nn_b=nn_b, q_t1=_f(q_sum) if with_residual else None,
                                q_t2=_f(q_fin))
    wi = w_idx.to(DEV)
    y, yi = be.linear_i8_nonorm(x_idx.to(DEV), wi, be.rowsum_i8(wi), bias.to(DEV), None if res is None else res.to(DEV),
                                nn_w.to(DEV), nn_b.to(DEV), _xq_dev(x_q), w_delta.to(DEV), 1e-8, _dev(q_dense),
                                _dev(q_sum) if with_residual else None, _dev(q_fin), torch.float32, want_idx=True)
    assert torch.equal(yi.cpu(), ref_i) and torch.equal(y.cpu(), ref_y)


@pytest.mark.gpu
def test_ffn_i8_equals_integer_oracle():
    from quantization import _hip
    be = _hip.backend()
    M, K1, N1, N2 = 1024, 128, 512, 128
    x_idx, w1, x_q, w1_delta, b1, q_mid = _rand_layer(M, N1, K1, seed=21, o_bits=4)
    q_mid = (q_mid[0] * 0.5, torch.tensor(0.0), None, 4, False, False, 1e-8)          # after ReLU: one-sided grid
    _, w2, _, w2_delta, b2, q_dense = _rand_layer(M, N2, N1, seed=22, x_bits=4, o_bits=4)
    nn_w, nn_b, q_sum, q_fin = _tail_args(N2, 23)
    res = torch.randn(M, N2, generator=torch.Generator().manual_seed(24))
    ref_y, ref_i = IO.ffn_i8(x_idx, x_q, w1, b1, w1_delta, 1e-8, _f(q_mid), w2, b2, w2_delta, 1e-8, res, nn_w, nn_b,
                             _f(q_dense), _f(q_sum), _f(q_fin))
    w1d, w2d = w1.to(DEV), w2.to(DEV)
    y, yi = be.ffn_i8_nonorm(x_idx.to(DEV), _xq_dev(x_q), w1d, be.rowsum_i8(w1d), b1.to(DEV), w1_delta.to(DEV), 1e-8,
                             _dev(q_mid), w2d, be.rowsum_i8(w2d), b2.to(DEV), w2_delta.to(DEV), 1e-8, res.to(DEV),
                             nn_w.to(DEV), nn_b.to(DEV), _dev(q_dense), _dev(q_sum), _dev(q_fin), torch.float32,
                             want_idx=True)
    assert torch.equal(yi.cpu(), ref_i) and torch.equal(y.cpu(), ref_y)


@pytest.mark.gpu
@pytest.mark.parametrize('scores_q', [True, False], ids=['scores-quantized', 'scores-fp32'])
@pytest.mark.parametrize('cfg', [(8, 128, 12, 64), (64, 128, 12, 64), (8, 128, 4, 32), (2, 64, 4, 32), (2, 256, 2, 64),
                                 (1, 512, 2, 64)], ids=lambda c: 'B%d-T%d-H%d-d%d' % c)
@pytest.mark.parametrize('split', ['auto', '0'], ids=['launch-auto', 'one-wave-rows'])
def test_attention_i8_equals_integer_oracle_bit_for_bit(cfg, scores_q, split, monkeypatch):
    """Scores, softmax (IEEE-only exponential, contractual summation tree) and both integer contractions: context
    values and indices equal the oracle's for the one-wave and the split-key launch shapes, with a padding mask."""
    from quantization import _hip
    be = _hip.backend()
    B, T, H, D = cfg
    if split != 'auto':
        if B > 8:
            pytest.skip('already the one-wave launch')
        monkeypatch.setenv('TQ_ATTN_SPLIT', split)          # read per call by tq_attention_i8_fwd
    g = torch.Generator().manual_seed(B * 1000 + T + H)
    qi, ki, vi = (torch.randint(-128, 128, (B, T, H * D), generator=g).to(torch.int8) for _ in range(3))
    if B > 8:                                            # keep the CPU side to seconds: check a slice of the batch
        sl = slice(0, 4)
    else:
        sl = slice(0, B)
    mask = torch.zeros(B, T)
    mask[0, T - 17:] = -10000.0
    if B > 1:
        mask[1, T // 2:] = -10000.0
    mk = lambda d, z, nb=8: (torch.tensor(d), torch.tensor(z), None, nb, False, False, 1e-8)
    q_q, q_k, q_v = mk(0.011, 120.0), mk(0.013, 131.0), mk(0.009, 128.0)
    q_s = mk(0.35, 128.0) if scores_q else None
    q_p, q_c = mk(1.0 / 255, 0.0), mk(0.012, 125.0, 4 if D == 32 else 8)
    if D == 32:
        q_c = mk(0.15, 8.0, 4)
    denom = float(np.sqrt(D))
    ctx, ci = be.attention_i8(qi.to(DEV), ki.to(DEV), vi.to(DEV), H, mask.to(DEV), denom, _dev(q_q), _dev(q_k), _dev(q_v),
                              _dev(q_s), _dev(q_p), _dev(q_c), want_idx=True)
    ref, ri = IO.attention_i8(qi[sl], ki[sl], vi[sl], H, mask[sl], denom, _f(q_q), _f(q_k), _f(q_v), _f(q_s), _f(q_p), _f(q_c))
    assert torch.equal(ci[sl].cpu(), ri)
    assert torch.equal(ctx[sl].cpu(), ref)
