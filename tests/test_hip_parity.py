"""GPU parity: the HIP path (through the drop-in classes, i.e. ctypes -> C ABI -> gfx950 kernels)
against the committed golden vectors and against the CPU oracle on seeded inputs.

Bars: integer indices bit-exact; dequantised floats bit-exact for fp32 (asserted) and within
1e-5 relative where transcendental functions are involved (AdaRound, cross-entropy).
"""
import numpy as np
import pytest
import torch

from oracle import tq_oracle as O
from tests._cases import fq_case, est_inputs, t, LAYOUT_ARGS

pytestmark = pytest.mark.gpu

DEV = 'cuda'


@pytest.fixture(scope='module')
def q():
    from quantization import _hip
    assert _hip.backend().name == 'hip'
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators, OptMethod
    from quantization.quantization_manager import QuantizationManager
    from utils.per_embd_quant_utils import set_act_quant_axis_and_groups
    import types
    return types.SimpleNamespace(**locals())


@pytest.fixture(params=[True, False], ids=['fused-calibration', 'layered-calibration'])
def calib_path(request):
    """Run calibration-state tests through both the one-call fused path (tq_calibrate_minmax) and
    the layered estimator -> set_quant_range -> quantizer path."""
    from quantization import quantization_manager as qm
    prev = qm.FUSED_CALIBRATION
    qm.FUSED_CALIBRATION = request.param
    yield request.param
    qm.FUSED_CALIBRATION = prev


def _manager(q, m, init='current_minmax', init_params=None):
    la = LAYOUT_ARGS[m['layout']]
    mgr = q.QuantizationManager(qmethod=q.QMethods[m['method']], init=q.RangeEstimators[init],
                                per_channel=la['per_channel'], qparams=dict(n_bits=m['n_bits']),
                                init_params=dict(init_params or {}))
    if la['axis'] is not None:
        q.set_act_quant_axis_and_groups(mgr, axis=la['axis'], n_groups=la['n_groups'],
                                        permute=m['layout'].endswith('_perm'))
    return mgr


def test_golden_fake_quant_through_manager(q, golden_fake_quant, calib_path):
    z, meta = golden_fake_quant
    for m in meta:
        c = fq_case(z, m)
        x = c['x'].to(DEV)
        mgr = _manager(q, m)
        if m['layout'].endswith('_perm'):
            assert mgr(x) is x                                   # phase 1 passes x through
            assert torch.equal(mgr.range_estimator.ranges.cpu(), c['ranges']), m
            mgr.range_estimator.per_group_range_estimation = False
        y = mgr(x)
        est, qz = mgr.range_estimator, mgr.quantizer
        assert torch.equal(est.current_xmin.cpu().reshape(-1), c['xmin'].reshape(-1)), m
        assert torch.equal(est.current_xmax.cpu().reshape(-1), c['xmax'].reshape(-1)), m
        assert torch.equal(qz._delta.cpu().reshape(-1), c['delta'].reshape(-1)), m
        if c['zero_float'] is not None:
            assert torch.equal(qz._zero_float.cpu().reshape(-1), c['zero_float'].reshape(-1)), m
        if c['symmetric']:
            assert qz.signed == m['signed']
            assert (float(qz.int_min), float(qz.int_max)) == (m['int_min'], m['int_max'])
        idx = qz.to_integer_forward(x)
        assert torch.equal(idx.cpu(), c['idx']), m
        if m['io'] == 'bf16':
            assert y.dtype == torch.bfloat16
            assert torch.equal(y.cpu(), c['y_bf16']), m
        else:
            assert torch.equal(y.cpu(), c['y']), m
        mgr.fix_ranges()
        assert torch.equal(mgr(x), y)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(1,), (7,), (3, 5, 24), (8, 128, 768), (2, 3, 4099), (64, 1000)])
def test_fake_quant_vs_oracle_per_tensor(q, dtype, shape):
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(*shape, generator=g) * 3).to(dtype)
    for sym, n_bits, rng in [(False, 8, (-2.5, 7.0)), (True, 8, (-3.0, 2.0)), (True, 4, (0.0, 5.0)),
                             (False, 4, (-1.0, 1.0)), (False, 16, (-9.0, 9.0))]:
        if sym:
            delta, signed = O.sym_params_from_range(*rng, n_bits)
            zf, sgn = None, bool(signed)
        else:
            delta, zf = O.asym_params_from_range(*rng, n_bits)
            signed, sgn = None, False
        ref_idx, ref_y = O.fake_quant_lowp(x, delta, zf, n_bits, sym, sgn)
        y, idx = be.fake_quant(x.to(DEV), delta.to(DEV), None if zf is None else zf.to(DEV),
                               None if signed is None else signed.to(DEV), n_bits, sym, False,
                               1e-8, 1, 1, idx_dtype=torch.int32)
        assert torch.equal(idx.cpu().float(), ref_idx), (dtype, shape, sym, n_bits)
        assert torch.equal(y.cpu(), ref_y), (dtype, shape, sym, n_bits)


def test_fake_quant_unaligned_and_index_dtypes(q):
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(5)
    base = (torch.randn(4099, generator=g) * 4).to(DEV)
    x = base[1:]                       # 4-byte aligned only -> scalar fallback kernel
    delta, zf = O.asym_params_from_range(-3.0, 5.0, 8)
    ref_idx, ref_y = O.fake_quant(x.cpu(), delta, zf, 8, False)
    for idt in (torch.float32, torch.uint8, torch.int16, torch.int32):
        y, idx = be.fake_quant(x, delta.to(DEV), zf.to(DEV), None, 8, False, False, 1e-8, 1, 1,
                               idx_dtype=idt)
        assert torch.equal(y.cpu(), ref_y)
        assert torch.equal(idx.cpu().float(), ref_idx)
    dlt, sg = O.sym_params_from_range(-3.0, 5.0, 8)
    ref_idx, _ = O.fake_quant(base.cpu(), dlt, None, 8, True, True)
    _, idx = be.fake_quant(base, dlt.to(DEV), None, sg.to(DEV), 8, True, False, 1e-8, 1, 1,
                           want_y=False, idx_dtype=torch.int8)
    assert torch.equal(idx.cpu().float(), ref_idx)


def _tie_adjacent(scale, ks, spread=3):
    """fp32 inputs whose quotient by `scale` sits on / next to the rounding ties k + 1/2: the cases where
    the kernels' reciprocal fast path must hand over to the true division (csrc/tq_device.h)."""
    import numpy as np
    base = ((ks.astype(np.float64) + 0.5) * np.float64(scale)).astype(np.float32)
    out = [base]
    lo, hi = base.copy(), base.copy()
    for _ in range(spread):
        lo = np.nextafter(lo, np.float32(-np.inf)); hi = np.nextafter(hi, np.float32(np.inf))
        out += [lo.copy(), hi.copy()]
    return np.concatenate(out)


@pytest.mark.parametrize('n_bits', [4, 8, 16])
def test_rounding_ties_are_bit_exact(q, n_bits):
    """rne(x / scale) through the guarded-reciprocal path == true division, on tie-adjacent inputs,
    for awkward scales (per-tensor, per-embedding and inside the MSE candidate kernel)."""
    import numpy as np
    from quantization import _hip
    be = _hip.backend()
    rs = np.random.RandomState(n_bits)
    top = 2 ** n_bits - 1
    ks = np.arange(-8, top + 8)
    if ks.size > 4096:
        ks = np.concatenate([ks[:600], rs.choice(ks, 3000, replace=False), ks[-600:]])
    scales = np.concatenate([[1.0, 0.1, 1.0 / 3.0, 0.7, 1e-8, 3.0000002, 0.049999997, 123.456],
                             np.exp(rs.uniform(-12, 6, 24))]).astype(np.float32)
    # per tensor, asymmetric + symmetric
    for sc in scales:
        x = torch.from_numpy(_tie_adjacent(sc, ks))
        x = torch.cat([x, -x])
        x = x[:x.numel() // 8 * 8]
        delta = torch.tensor(float(sc))
        zf = torch.tensor(float(rs.uniform(0, top)))
        ref_idx, ref_y = O.fake_quant(x, delta, zf, n_bits, False)
        y, idx = be.fake_quant(x.to(DEV), delta.to(DEV), zf.to(DEV), None, n_bits, False, False, 1e-8, 1, 1,
                               idx_dtype=torch.int32)
        assert torch.equal(idx.cpu().float(), ref_idx), float(sc)
        assert torch.equal(y.cpu(), ref_y), float(sc)
        sg = torch.tensor(True)
        ref_idx, ref_y = O.fake_quant(x, delta, None, n_bits, True, True)
        y, idx = be.fake_quant(x.to(DEV), delta.to(DEV), None, sg.to(DEV), n_bits, True, False, 1e-8, 1, 1,
                               idx_dtype=torch.int32)
        assert torch.equal(idx.cpu().float(), ref_idx) and torch.equal(y.cpu(), ref_y), float(sc)
    # per embedding: column c has its own scale; rows = tie-adjacent values of that column
    d = 32
    sc = scales[:d]
    kk = ks[:: max(1, ks.size // 200)]
    cols = [_tie_adjacent(s_, kk) for s_ in sc]
    x = torch.from_numpy(np.stack(cols, axis=1).copy())             # [rows, d]
    delta = torch.from_numpy(sc.copy())
    zf = torch.from_numpy(rs.uniform(0, top, d).astype(np.float32))
    ref_idx, ref_y = O.fake_quant(x, delta, zf, n_bits, False, axis=1)
    y, idx = be.fake_quant(x.to(DEV), delta.to(DEV), zf.to(DEV), None, n_bits, False, False, 1e-8, d, 1,
                           idx_dtype=torch.int32)
    assert torch.equal(idx.cpu().float(), ref_idx) and torch.equal(y.cpu(), ref_y)
    xb = x.to(torch.bfloat16)
    ref_idx, ref_y = O.fake_quant_lowp(xb, delta, zf, n_bits, False, axis=1)
    y, idx = be.fake_quant(xb.to(DEV), delta.to(DEV), zf.to(DEV), None, n_bits, False, False, 1e-8, d, 1,
                           idx_dtype=torch.int32)
    assert torch.equal(idx.cpu().float(), ref_idx) and torch.equal(y.cpu(), ref_y)


def test_mse_candidate_losses_on_tie_adjacent_inputs(q):
    """Candidate kernel: a single mis-rounded element changes a loss by ~scale^2; compare against fp64
    sums of the oracle's per-element errors."""
    import numpy as np
    from quantization import _hip
    be = _hip.backend()
    rs = np.random.RandomState(3)
    scales = np.exp(rs.uniform(-6, 1, 48)).astype(np.float32)
    ks = np.arange(0, 256)
    x = torch.from_numpy(np.concatenate([_tie_adjacent(s_, ks, spread=1) for s_ in scales[:8]]))
    x = x[:x.numel() // 1024 * 1024].contiguous()
    tab = np.stack([scales, rs.randint(0, 255, scales.size).astype(np.float32),
                    np.zeros_like(scales), np.full_like(scales, 255.0)], axis=1).astype(np.float32)
    ref = []
    for sc_, zp_, lo_, hi_ in tab:
        xi = torch.clamp(torch.round(x / float(sc_)) + float(zp_), float(lo_), float(hi_))
        err = x - float(sc_) * (xi - float(zp_))
        ref.append(float((err.double() ** 2).sum()))
    loss = be.zeros_f64((1, scales.size), DEV)
    be.mse_candidates(x.to(DEV), 1, torch.from_numpy(tab).to(DEV), loss)
    got = loss.cpu().numpy()[0]
    assert np.allclose(got, np.array(ref), rtol=2e-6, atol=0), np.abs(got / np.array(ref) - 1).max()


def test_special_values_nan_inf(q):
    from quantization import _hip
    x = torch.tensor([float('nan'), float('inf'), -float('inf'), 0.0, -0.0, 1e-30, 0.5, 1.5, 2.5,
                      -0.5, 3e38], dtype=torch.float32)
    delta, zf = O.asym_params_from_range(-4.0, 4.0, 8)
    ref_idx, ref_y = O.fake_quant(x, delta, zf, 8, False)
    y, idx = _hip.backend().fake_quant(x.to(DEV), delta.to(DEV), zf.to(DEV), None, 8, False, False,
                                       1e-8, 1, 1, idx_dtype=torch.float32)
    assert torch.equal(torch.isnan(idx.cpu()), torch.isnan(ref_idx))
    ok = ~torch.isnan(ref_idx)
    assert torch.equal(idx.cpu()[ok], ref_idx[ok])
    assert torch.equal(y.cpu()[ok], ref_y[ok])
    # bit patterns, i.e. the sign of zero too (x = -0.0 dequantises to +0.0 in the reference: x_int - zero_point)
    assert torch.equal(y.cpu()[ok].view(torch.int32), ref_y[ok].view(torch.int32))
    assert torch.equal(idx.cpu()[ok].view(torch.int32), ref_idx[ok].view(torch.int32))
    # symmetric grid (zero_point = 0): round(-0.0) + 0.0 = +0.0 upstream
    ds, sg = O.sym_params_from_range(-4.0, 4.0, 8)
    ref_idx, ref_y = O.fake_quant(x, ds, None, 8, True, True)
    y, idx = _hip.backend().fake_quant(x.to(DEV), ds.to(DEV), None, sg.to(DEV), 8, True, False, 1e-8, 1, 1,
                                       idx_dtype=torch.float32)
    assert torch.equal(y.cpu()[ok].view(torch.int32), ref_y[ok].view(torch.int32))
    assert torch.equal(idx.cpu()[ok].view(torch.int32), ref_idx[ok].view(torch.int32))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('n_bits', [8, 4, 2])
def test_byte_index_outputs_equal_the_float_indices(q, dtype, n_bits):
    """uint8 indices and int8(index - 128) (the integer GEMM's operand) for grids inside [0, 255] are packed with one
    v_cvt_pk_u8_f32 per element (csrc/tq_fake_quant.hip store_idx): equal to the float index tensor converted element by
    element, NaN inputs give byte 0 / -128 (the conversion's value for NaN, as before), +-inf the grid ends; per-tensor,
    per-embedding and per-token launches, with and without the dequantised output."""
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(n_bits)
    x = (torch.randn(6, 40, 64, generator=g) * 3).to(dtype)
    x.view(-1)[[0, 77, 4095, 9000]] = torch.tensor([float('nan'), float('inf'), -float('inf'), float('nan')]).to(dtype)
    xd = x.to(DEV)
    for n_params, inner in ((1, 1), (64, 1), (40, 64)):
        lo = -torch.rand(n_params, generator=g) * 4 - 1
        hi = torch.rand(n_params, generator=g) * 4 + 1
        delta, zf = O.asym_params_from_range(lo, hi, n_bits)
        args = (delta.to(DEV), zf.to(DEV), None, n_bits, False, False, 1e-8, n_params, inner)
        y, idx = be.fake_quant(xd, *args, idx_dtype=torch.float32)
        want = torch.nan_to_num(idx, nan=0.0)
        assert float(want.min()) >= 0 and float(want.max()) <= 2 ** n_bits - 1
        for want_y in (True, False):
            y8, i8 = be.fake_quant(xd, *args, want_y=want_y, idx_dtype=torch.uint8)
            assert i8.dtype == torch.uint8 and torch.equal(i8.float(), want), (n_params, want_y)
            if want_y:
                assert torch.equal(y8.view(torch.int16 if y8.element_size() == 2 else torch.int32),
                                   y.view(torch.int16 if y.element_size() == 2 else torch.int32))
        if n_params == 1:
            _, im = be.fake_quant_int8(xd, delta.to(DEV), zf.to(DEV), n_bits, 1e-8)
            assert im.dtype == torch.int8 and torch.equal(im.float(), want - 128)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape,axis', [((8, 128, 768), 2), ((4, 16, 3072), 2), ((8, 768), 1),
                                        ((5, 7, 24), 2), ((3, 40, 6), 1), ((6, 10), 0)])
def test_minmax_and_axis_quant_vs_oracle(q, dtype, shape, axis):
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(shape[-1] + axis)
    x = (torch.randn(*shape, generator=g) * torch.linspace(0.5, 4, shape[-1])).to(dtype)
    xf = x.float()
    inner = int(np.prod(shape[axis + 1:]))
    mn, mx = be.minmax(x.to(DEV), shape[axis], inner)
    rmn, rmx = O.minmax_axis(xf, axis)
    assert torch.equal(mn.cpu(), rmn) and torch.equal(mx.cpu(), rmx)
    tmn, tmx = be.minmax(x.to(DEV), 1, 1)
    assert float(tmn) == float(xf.min()) and float(tmx) == float(xf.max())
    delta, zf = O.asym_params_from_range(rmn, rmx, 8)
    d2, z2 = be.set_range_asym(mn, mx, 8, 1e-8, False)
    assert torch.equal(d2.cpu(), delta) and torch.equal(z2.cpu(), zf)
    ref_idx, ref_y = O.fake_quant_lowp(x, delta, zf, 8, False, axis=axis)
    y, idx = be.fake_quant(x.to(DEV), d2, z2, None, 8, False, False, 1e-8, shape[axis], inner,
                           idx_dtype=torch.float32)
    assert torch.equal(idx.cpu(), ref_idx)
    assert torch.equal(y.cpu(), ref_y)


def _check_trace(q, m, z):
    k = m['k']
    xs = [x.to(DEV) for x in est_inputs(z, m)]
    ip = dict(m['init_params'])
    golden_section = ip.get('opt_method') == 'golden_section'
    if 'opt_method' in ip:
        ip['opt_method'] = q.OptMethod[ip['opt_method']]
    mgr = _manager(q, m, init=m['init'], init_params=ip)
    for b, x in enumerate(xs):
        y = mgr(x)
        est = mgr.range_estimator
        got_min = est.current_xmin.cpu().reshape(-1)
        got_max = est.current_xmax.cpu().reshape(-1)
        ref_min, ref_max = t(z[f'e{k}_xmin'][b]), t(z[f'e{k}_xmax'][b])
        if m['init'] == 'cross_entropy':
            assert torch.allclose(got_max, ref_max, rtol=1e-6), (m, b)
            assert torch.allclose(got_min, ref_min, rtol=1e-6), (m, b)
        else:
            assert torch.equal(got_min, ref_min), (m, b, got_min, ref_min)
            assert torch.equal(got_max, ref_max), (m, b, got_max, ref_max)
            assert torch.equal(mgr.quantizer._delta.cpu().reshape(-1), t(z[f'e{k}_delta'][b])), (m, b)
    if m['init'] != 'cross_entropy':
        assert torch.equal(y.cpu(), t(z[f'e{k}_y_last'])), m
    la = getattr(mgr.range_estimator, 'loss_array', None)
    if la is not None and f'e{k}_loss_array' in z.files and not golden_section:
        ref = z[f'e{k}_loss_array']
        fin = np.isfinite(ref)
        if m['init'] == 'MSE':
            # every candidate loss is the reference's fp32 torch.sum value, accumulated over batches in fp64
            # like the reference's numpy loss_array: equal bits, not a tolerance
            assert np.array_equal(la[fin], ref[fin]), (m, np.abs(la[fin] / ref[fin] - 1).max())
        else:
            assert np.allclose(la[fin], ref[fin], rtol=2e-5, atol=1e-6), m
        assert np.array_equal(np.isfinite(la), fin)


def test_estimator_traces(q, golden_estimators, calib_path):
    z, meta = golden_estimators
    for m in meta:
        _check_trace(q, m, z)


def test_permuted_peg_trace(q, golden_estimators, calib_path):
    z, _ = golden_estimators
    xs = [t(b).to(DEV) for b in z['batches']]
    mgr = q.QuantizationManager(qmethod=q.QMethods.asymmetric_uniform,
                                init=q.RangeEstimators.current_minmax, qparams=dict(n_bits=8))
    q.set_act_quant_axis_and_groups(mgr, axis=2, n_groups=4, permute=True)
    for x in xs:
        assert mgr(x) is x
    assert torch.equal(mgr.range_estimator.ranges.cpu(), t(z['perm_ranges']))
    mgr.range_estimator.per_group_range_estimation = False
    y = mgr(xs[0])
    assert torch.equal(mgr.range_estimator.current_xmin.cpu(), t(z['perm_xmin']))
    assert torch.equal(mgr.range_estimator.current_xmax.cpu(), t(z['perm_xmax']))
    assert torch.equal(y.cpu(), t(z['perm_y']))


def test_mse_losses_vs_oracle_large_grid(q):
    """1-D (101 candidates) and 2-D (8-bit: 20 x 64 x 2) searches on a BERT-shaped tensor:
    same argmin as the oracle's per-candidate loop, losses within 1e-5 relative."""
    g = torch.Generator().manual_seed(77)
    x = torch.randn(4, 32, 96, generator=g)
    x[..., 7] *= 15
    for method, n_bits, C in (('symmetric_uniform', 8, 100), ('asymmetric_uniform', 8, 6),
                              ('asymmetric_uniform', 4, 30)):
        m = dict(method=method, n_bits=n_bits, layout='per_tensor')
        mgr = _manager(q, m, init='MSE', init_params=dict(num_candidates=C))
        mgr(x.to(DEV))
        qs = O.QSpec(n_bits, method == 'symmetric_uniform')
        s = O.MSESearch(qs, num_candidates=C)
        rmin, rmax = s.step_batch(x)
        got = mgr.range_estimator.loss_array
        fin = np.isfinite(s.loss_array)
        assert np.allclose(got[fin], s.loss_array[fin], rtol=1e-5), method
        assert torch.equal(mgr.range_estimator.current_xmin.cpu(), rmin), method
        assert torch.equal(mgr.range_estimator.current_xmax.cpu(), rmax), method


def test_ste_backward_matches_autograd(q):
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(33, 65, generator=g) * 3)
    go = torch.randn(33, 65, generator=g)
    for method in ('asymmetric_uniform', 'symmetric_uniform'):
        mgr = q.QuantizationManager(qmethod=q.QMethods[method], qparams=dict(n_bits=4),
                                    x_min=-2.0, x_max=2.5)
        qz = mgr.quantizer
        sym = method == 'symmetric_uniform'
        ref_y, ref_dx, ref_dd, ref_dz = O.fake_quant_with_grads(
            x, qz._delta.cpu(), None if sym else qz._zero_float.cpu(), 4, sym,
            signed=(qz.signed if sym else False), grad_out=go)
        xd = x.to(DEV).requires_grad_(True)
        mgr.learn_ranges()
        y = mgr(xd)
        y.backward(go.to(DEV))
        assert torch.equal(y.detach().cpu(), ref_y)
        assert torch.allclose(xd.grad.cpu(), ref_dx, rtol=1e-6, atol=1e-7)
        assert torch.allclose(qz._delta.grad.cpu(), ref_dd, rtol=1e-4, atol=1e-3), method
        if not sym:
            assert torch.allclose(qz._zero_float.grad.cpu(), ref_dz, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('symmetric', [False, True], ids=['asym', 'sym'])
def test_ste_backward_log_scale_domain(symmetric):
    """scale_domain='log' (scale = exp(delta), reference quantizers.py:142-147): tq_fake_quant_bwd's range gradients
    d/d log(delta) against autograd through the oracle's op chain.  expf on the device and torch's CPU exp agree to
    1 ulp only, so an index can flip where x / scale sits on a rounding tie: the element gradient is compared where the
    forward index agrees (all but a handful), the parameter gradients at the tolerance of the linear-domain test."""
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(31)
    x = torch.randn(48, 96, generator=g) * 2.5
    go = torch.randn(48, 96, generator=g)
    log_delta = torch.log(torch.tensor(0.21))
    zf = None if symmetric else torch.tensor(7.3)
    sgn = torch.tensor(True) if symmetric else None
    ref_y, ref_dx, ref_dd, ref_dz = O.fake_quant_with_grads(x, log_delta, zf, 4, symmetric, signed=True, grad_out=go,
                                                            scale_domain='log')
    y, _ = be.fake_quant(x.to(DEV), log_delta.to(DEV), None if zf is None else zf.to(DEV), None if sgn is None else
                         sgn.to(DEV), 4, symmetric, True, 1e-8, 1, 1)
    same = (y.cpu() - ref_y).abs() <= 1e-6 * ref_y.abs().clamp(min=1.0)
    assert float(same.float().mean()) >= 0.999
    gx, gd, gz = be.fake_quant_bwd(x.to(DEV), go.to(DEV), log_delta.to(DEV), None if zf is None else zf.to(DEV),
                                   None if sgn is None else sgn.to(DEV), 4, symmetric, True, 1e-8, 1, 1, param_grads=True)
    assert torch.allclose(gx.cpu()[same], ref_dx[same], rtol=1e-6, atol=1e-7)
    assert torch.allclose(gd.cpu().reshape(()), ref_dd, rtol=2e-3, atol=2e-2), (float(gd), float(ref_dd))
    if not symmetric:
        assert torch.allclose(gz.cpu().reshape(()), ref_dz, rtol=2e-3, atol=2e-2), (float(gz), float(ref_dz))


# ------------------------------------------------------------------------------------------
# size-independent properties at BASELINE.json's full size: [B, S, 768] bf16 hidden states
# ------------------------------------------------------------------------------------------
def test_full_size_properties(q, calib_path):
    B, S, D = 1024, 512, 768
    torch.manual_seed(1000)
    x = torch.randn(B, S, D, device=DEV, dtype=torch.bfloat16)
    x[..., 308] *= 20
    x[..., 381] *= 20
    mgr = q.QuantizationManager(qmethod=q.QMethods.asymmetric_uniform,
                                init=q.RangeEstimators.running_minmax, qparams=dict(n_bits=8))
    y = mgr(x)
    est, qz = mgr.range_estimator, mgr.quantizer
    # statistics equal torch's own reductions
    assert float(est.current_xmin) == float(x.min()) and float(est.current_xmax) == float(x.max())
    mgr.fix_ranges()
    # deterministic + idempotent: Q(Q(x)) == Q(x) once the grid is fixed
    assert torch.equal(mgr(x), y)
    assert torch.equal(mgr(y), y)
    # indices live on the grid and reproduce y when dequantised
    idx = qz.to_integer_forward(x)
    assert float(idx.min()) >= 0 and float(idx.max()) <= 255
    assert torch.equal(idx, torch.round(idx))
    deq = (qz.scale * (idx - qz.zero_point)).to(torch.bfloat16)
    assert torch.equal(deq, y)
    # a slab of the big tensor agrees bit-for-bit with the CPU oracle
    sl = x[5, :64].cpu()
    ref_idx, ref_y = O.fake_quant_lowp(sl, qz._delta.cpu(), qz._zero_float.cpu(), 8, False)
    assert torch.equal(idx[5, :64].cpu(), ref_idx)
    assert torch.equal(y[5, :64].cpu(), ref_y)
    # per-embedding statistics at full size: column-owned kernel == torch reductions
    from quantization import _hip
    mn, mx = _hip.backend().minmax(x, D, 1)
    assert torch.equal(mn, x.view(-1, D).float().amin(0)) and torch.equal(mx, x.view(-1, D).float().amax(0))


def test_empty_tensor(q):
    from quantization import _hip
    be = _hip.backend()
    delta, zf = O.asym_params_from_range(-1.0, 1.0, 8)
    y, _ = be.fake_quant(torch.empty(0, 768, device=DEV), delta.to(DEV), zf.to(DEV), None, 8, False,
                         False, 1e-8, 1, 1)
    assert y.shape == (0, 768)
    with pytest.raises(_hip.TQError):
        be.minmax(torch.empty(0, device=DEV), 1, 1)     # like torch.min of an empty tensor


def test_cpu_tensor_is_refused(q):
    from quantization import _hip
    mgr = q.QuantizationManager(qmethod=q.QMethods.asymmetric_uniform, qparams=dict(n_bits=8))
    with pytest.raises(_hip.TQError):
        mgr(torch.randn(4, 4))


# ------------------------------------------------------------------------------------------
# AdaRound kernels (K10 / K11 / K13) -- floating point with exp/log: 1e-5 relative
# ------------------------------------------------------------------------------------------
def _ada_layer(q, z, m):
    from quantization.autoquant_utils import QuantLinear
    from quantization.adaround.quantizer import ADAROUND_QUANTIZER_MAP
    from quantization.adaround.utils import AdaRoundMode
    k = m['k']
    layer = QuantLinear(16, 12, method=q.QMethods[m['method']], n_bits=4,
                        weight_range_method=q.RangeEstimators.current_minmax)
    layer.weight.data = t(z[f'a{k}_w']).clone()
    layer.bias.data = t(z[f'a{k}_b']).clone()
    layer = layer.to(DEV)
    layer.quantized_weights()
    layer.caching = False
    X, tgt = t(z[f'a{k}_X']).to(DEV), t(z[f'a{k}_tgt']).to(DEV)
    with torch.no_grad():
        layer(X[:4])
    oq = layer.weight_quantizer.quantizer
    wq = ADAROUND_QUANTIZER_MAP[oq.__class__](n_bits=oq.n_bits, scale_domain=oq.scale_domain,
                                              per_channel=oq.per_channel, eps=oq.eps)
    for name in ('_delta', '_zero_float', '_signed'):
        if hasattr(oq, name):
            wq.register_buffer(name, getattr(oq, name))
    layer.weight_quantizer.quantizer = wq
    layer.weight_quantizer.fix_ranges()
    wq.round_mode = AdaRoundMode[m['mode']]
    wq.temperature = 20
    return layer, wq, X, tgt


def test_adaround_kernels_vs_golden(q, golden_adaround):
    """alpha initialisation, soft and hard forward against the reference's recorded tensors."""
    z, meta = golden_adaround
    for m in meta:
        k = m['k']
        layer, wq, X, tgt = _ada_layer(q, z, m)
        assert torch.equal(wq._delta.cpu(), t(z[f'a{k}_delta'])), m
        wq.soft_targets = True
        with torch.no_grad():
            soft0 = wq(layer.weight)
        assert torch.allclose(wq.alpha.detach().cpu(), t(z[f'a{k}_alpha0']), rtol=1e-5, atol=1e-5), m
        assert torch.allclose(soft0.cpu(), t(z[f'a{k}_wq_soft0']), rtol=1e-5, atol=1e-6), m
        wq.soft_targets = False
        with torch.no_grad():
            hard0 = wq(layer.weight)
            idx0 = wq.to_integer_forward(layer.weight)
        assert torch.equal(hard0.cpu(), t(z[f'a{k}_wq_hard0'])), m        # hard rounding: exact
        assert torch.allclose(idx0.cpu(), t(z[f'a{k}_idx_hard0']), atol=1e-4), m
        # hard rounding with the reference's FINAL alpha reproduces its final weights exactly
        wq.alpha.data = t(z[f'a{k}_alphas'][-1]).to(DEV)
        with torch.no_grad():
            assert torch.equal(wq(layer.weight).cpu(), t(z[f'a{k}_wq_hard1'])), m


@pytest.mark.parametrize('mode', ['learned_hard_sigmoid', 'learned_sigmoid', 'sigmoid_temp_decay'])
@pytest.mark.parametrize('method', ['symmetric_uniform', 'asymmetric_uniform'])
def test_adaround_fused_step_vs_torch_adam(q, mode, method):
    """K11 (gradient through K10 + regulariser + Adam) against autograd + torch.optim.Adam on the
    CPU oracle, driven by the SAME well-conditioned weight gradients for several steps (the GEMM is
    outside the kernel under test).  Also checks the stand-alone backward kernel."""
    from quantization import _hip
    be = _hip.backend()
    mcode = {'learned_sigmoid': 0, 'learned_hard_sigmoid': 1, 'sigmoid_temp_decay': 2}[mode]
    sym = method == 'symmetric_uniform'
    g = torch.Generator().manual_seed(17 + mcode)
    w = torch.randn(96, 64, generator=g) * 0.1
    if sym:
        delta, signed = O.sym_params_from_range(w.min(), w.max(), 4)
        zf, sgn = None, bool(signed)
    else:
        delta, zf = O.asym_params_from_range(w.min(), w.max(), 4)
        signed, sgn = None, False
    temp = 7.0
    alpha0 = O.ada_alpha_init(w, O.effective_scale(delta), mode, temp) + torch.randn(96, 64, generator=g)
    qargs = (delta.to(DEV), None if zf is None else zf.to(DEV), None if signed is None else signed.to(DEV),
             4, sym, False, 1e-8, 1, 1)
    # reference: autograd + torch.optim.Adam
    a_ref = alpha0.clone().requires_grad_(True)
    opt = torch.optim.Adam([a_ref], lr=1e-2)
    a_dev = alpha0.clone().to(DEV)
    m_dev, v_dev = torch.zeros_like(a_dev), torch.zeros_like(a_dev)
    wd = w.to(DEV)
    for step in range(1, 6):
        gw = torch.randn(96, 64, generator=g)
        beta, reg_w = 20.0 - 3.0 * step, (0.0 if step == 1 else 0.01)
        opt.zero_grad()
        _, wq_ref = O.ada_fake_quant(w, a_ref, delta, zf, 4, sym, sgn, mode, True, temperature=temp)
        obj = (wq_ref * gw).sum()
        if reg_w:
            obj = obj + O.ada_round_reg(a_ref, mode, beta, reg_w, temp)
        obj.backward()
        g_ref = a_ref.grad.clone()
        if reg_w == 0.0:
            g_alone = be.adaround_bwd(wd, a_dev, gw.to(DEV), qargs, mcode, temp)
            assert torch.allclose(g_alone.cpu(), g_ref, rtol=1e-4, atol=1e-7), (mode, method, step)
        wq_dev = be.adaround_fwd(wd, a_dev, qargs, mcode, True, temp)
        assert torch.allclose(wq_dev.cpu(), wq_ref.detach(), rtol=1e-5, atol=1e-6)
        g_dev = be.adaround_bwd_adam(wd, gw.to(DEV), a_dev, m_dev, v_dev, qargs, mcode, temp, reg_w,
                                     beta, 1e-2, 0.9, 0.999, 1e-8, step, want_grad=True)
        opt.step()
        assert torch.allclose(g_dev.cpu(), g_ref, rtol=2e-4, atol=1e-6), (mode, method, step)
        assert torch.allclose(a_dev.cpu(), a_ref.detach(), rtol=1e-4, atol=2e-5), (mode, method, step)


@pytest.mark.parametrize('beta', [2.0, 1.0, 0.5])
def test_adaround_regulariser_gradient_at_the_kink(beta):
    """the fused step's regulariser gradient where h(alpha) == 0.5 exactly (u = |2h - 1| = 0; alpha = 0 for
    both sigmoid modes).  d/dalpha [1 - u^beta] there is 0 in torch for beta >= 1 (abs'(0) = sign(0) = 0) -- the kernel
    agrees; for beta < 1 torch's chain rule gives 0 * inf = NaN (outside the reference's annealing range 20 -> 2,
    adaround/config.py) and the kernel is DEFINED to return 0: a finite step instead of a NaN that would poison alpha."""
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(5)
    w = torch.randn(8, 64, generator=g) * 0.1
    delta, signed = O.sym_params_from_range(w.min(), w.max(), 4)
    qargs = (delta.to(DEV), None, signed.to(DEV), 4, True, False, 1e-8, 1, 1)
    for mode, mcode in (('learned_sigmoid', 0), ('learned_hard_sigmoid', 1)):
        alpha0 = torch.randn(8, 64, generator=g)
        alpha0[:, ::4] = 0.0                                     # h == 0.5 exactly on a quarter of the entries
        gw = torch.randn(8, 64, generator=g)
        a_ref = alpha0.clone().requires_grad_(True)
        _, wq_ref = O.ada_fake_quant(w, a_ref, delta, None, 4, True, bool(signed), mode, True, temperature=7.0)
        obj = (wq_ref * gw).sum() + O.ada_round_reg(a_ref, mode, beta, 0.01, 7.0)
        obj.backward()
        g_ref = a_ref.grad
        a_dev = alpha0.clone().to(DEV)
        m_dev, v_dev = torch.zeros_like(a_dev), torch.zeros_like(a_dev)
        g_dev = be.adaround_bwd_adam(w.to(DEV), gw.to(DEV), a_dev, m_dev, v_dev, qargs, mcode, 7.0, 0.01, beta, 1e-2, 0.9,
                                     0.999, 1e-8, 1, want_grad=True).cpu()
        kink = torch.zeros(8, 64, dtype=torch.bool)
        kink[:, ::4] = True
        assert torch.isfinite(g_dev).all() and torch.isfinite(a_dev).all(), (mode, beta)
        assert torch.allclose(g_dev[~kink], g_ref[~kink], rtol=2e-4, atol=1e-6), (mode, beta)
        # at the kink only the reconstruction term contributes
        a2 = alpha0.clone().requires_grad_(True)
        _, wq2 = O.ada_fake_quant(w, a2, delta, None, 4, True, bool(signed), mode, True, temperature=7.0)
        (wq2 * gw).sum().backward()
        assert torch.allclose(g_dev[kink], a2.grad[kink], rtol=2e-4, atol=1e-6), (mode, beta)
        if beta >= 1.0:
            assert torch.allclose(g_dev[kink], g_ref[kink], rtol=2e-4, atol=1e-6), (mode, beta)
        else:
            assert torch.isnan(g_ref[kink]).all()                # what the kernel deliberately does not reproduce


def test_adaround_layer_gradient_well_conditioned(q):
    """Class path (QuantLinear + AdaRoundQuantizer + autograd through _AdaRoundFn) against the CPU
    oracle on a problem whose loss is O(1) (so that GEMM round-off does not dominate)."""
    from quantization.autoquant_utils import QuantLinear
    from quantization.adaround.quantizer import ADAROUND_QUANTIZER_MAP
    from quantization.adaround.utils import AdaRoundMode
    g = torch.Generator().manual_seed(23)
    lin_w, lin_b = torch.randn(48, 32, generator=g) * 0.2, torch.randn(48, generator=g) * 0.1
    X, tgt = torch.randn(8, 10, 32, generator=g), torch.randn(8, 10, 48, generator=g)
    layer = QuantLinear(32, 48, method=q.QMethods.symmetric_uniform, n_bits=4)
    layer.weight.data, layer.bias.data = lin_w.clone(), lin_b.clone()
    layer = layer.to(DEV)
    layer.quantized_weights()
    layer.caching = False
    with torch.no_grad():
        layer(X.to(DEV))
    oq = layer.weight_quantizer.quantizer
    wq = ADAROUND_QUANTIZER_MAP[oq.__class__](n_bits=4)
    for name in ('_delta', '_zero_float', '_signed'):
        wq.register_buffer(name, getattr(oq, name))
    layer.weight_quantizer.quantizer = wq
    layer.weight_quantizer.fix_ranges()
    wq.round_mode = AdaRoundMode.learned_hard_sigmoid
    wq.soft_targets = True
    out = layer(X.to(DEV))
    loss = torch.nn.functional.mse_loss(out, tgt.to(DEV), reduction='none').sum(1).mean()
    loss.backward()
    # oracle
    delta, signed = O.sym_params_from_range(lin_w.min(), lin_w.max(), 4)
    a = O.ada_alpha_init(lin_w, O.effective_scale(delta), 'learned_hard_sigmoid').requires_grad_(True)
    _, wq_ref = O.ada_fake_quant(lin_w, a, delta, None, 4, True, bool(signed), 'learned_hard_sigmoid', True)
    ref_loss = O.ada_rec_loss(torch.nn.functional.linear(X, wq_ref, lin_b), tgt)
    ref_loss.backward()
    assert abs(float(loss) - float(ref_loss)) <= 1e-5 * abs(float(ref_loss))
    assert torch.allclose(wq.alpha.grad.cpu(), a.grad, rtol=2e-3, atol=1e-5)
    from quantization import _hip
    rec = _hip.backend().recon_loss(out.detach(), tgt.to(DEV))
    assert abs(float(rec) - float(ref_loss)) <= 1e-5 * abs(float(ref_loss))


def test_adaround_reg_and_recon_vs_oracle(q):
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(3)
    alpha = torch.randn(3072, 768, generator=g) * 4
    for mode, name in ((0, 'learned_sigmoid'), (1, 'learned_hard_sigmoid'), (2, 'sigmoid_temp_decay')):
        for beta in (20.0, 7.3, 2.0):
            ref = float(O.ada_round_reg(alpha, name, beta, 0.01, temperature=5.0))
            got = float(be.adaround_reg(alpha.to(DEV), mode, 5.0, beta, 0.01))
            assert abs(got - ref) <= 2e-5 * abs(ref) + 1e-6, (name, beta, got, ref)
    pred = torch.randn(8, 128, 768, generator=g)
    tgt = torch.randn(8, 128, 768, generator=g)
    ref = float(O.ada_rec_loss(pred, tgt))
    got = float(be.recon_loss(pred.to(DEV), tgt.to(DEV)))
    assert abs(got - ref) <= 1e-5 * abs(ref)


def test_apply_adaround_to_layer_end_to_end(q):
    """The reference's entry point on a BERT-shaped layer: the learned rounding must not be worse
    than nearest rounding on the layer's own reconstruction loss."""
    from quantization.base_quantized_model import QuantizedModel
    from quantization.autoquant_utils import quantize_model
    from quantization.adaround import apply_adaround_to_layer
    from quantization.adaround.config import DEFAULT_ADAROUND_CONFIG
    import copy
    torch.manual_seed(11)

    class Net(QuantizedModel):
        def __init__(self):
            super().__init__()
            qp = dict(method=q.QMethods.symmetric_uniform, n_bits=4,
                      act_method=q.QMethods.asymmetric_uniform, n_bits_act=8)
            self.fc1 = quantize_model(torch.nn.Linear(96, 192), **qp)
            self.fc2 = quantize_model(torch.nn.Linear(192, 96), **qp)

        def forward(self, x):
            return self.fc2(torch.nn.functional.gelu(self.fc1(x)))

    model = Net().to(DEV)
    data = torch.randn(64, 16, 96, device=DEV)
    model.set_quant_state(True, False)
    model.eval()
    with torch.no_grad():
        model(data[:8])           # initialise the weight ranges
    cfg = copy.deepcopy(DEFAULT_ADAROUND_CONFIG)
    cfg.iters = 200
    for layer in (model.fc1, model.fc2):
        model.full_precision()
        layer.quantized_weights()
        res = apply_adaround_to_layer(model, layer, data, batch_size=8, act_quant=False,
                                      adaround_config=cfg)
        assert res.loss_hard_after <= res.loss_hard_before * 1.0001, res
        if layer is model.fc1:
            assert res.loss_soft_before < 1e-8   # h(alpha0) == frac(w/s): FP32 weights exactly
        hard = layer.weight_quantizer.quantizer
        assert hard.soft_targets is False and layer.caching is True


def test_toy_model_calibration_on_gpu(q, golden_toy, calib_path):
    """pass_data_for_range_estimation on the GPU: weight-side state is exact; activation ranges
    depend on GEMM outputs (hipBLASLt vs CPU) and are compared at 1e-4 relative."""
    import json
    from tests.test_host_logic import ToyNet, _quant_toy
    from utils.utils import pass_data_for_range_estimation
    z, _ = golden_toy
    org = ToyNet()
    org.load_state_dict({k[2:]: t(z[k]) for k in z.files if k.startswith('w_')})
    qp = dict(method=q.QMethods.symmetric_uniform, act_method=q.QMethods.asymmetric_uniform,
              n_bits=8, n_bits_act=8, weight_range_method=q.RangeEstimators.current_minmax,
              act_range_method=q.RangeEstimators.running_minmax)
    model = _quant_toy(org, **qp).to(DEV)
    loader = [(t(b),) for b in z['loader']]
    pass_data_for_range_estimation(loader, model, act_quant=True, weight_quant=True, max_num_batches=3)
    model.fix_ranges()
    model.eval()
    out = model(loader[3][0].to(DEV))
    sd = model.state_dict()
    for name in json.loads(str(z['sd_names'])):
        ref = torch.from_numpy(z['sd_' + name])
        got = sd[name].cpu()
        assert got.shape == ref.shape, name
        if 'weight_quantizer' in name:
            assert torch.equal(got.to(ref.dtype), ref), name
        else:
            assert torch.allclose(got.float(), ref.float(), rtol=1e-4, atol=1e-5), name
    # outputs sit on an 8-bit grid: allow one quantisation step of the final quantizer
    step = float(sd['res_q.activation_quantizer.quantizer._delta'])
    assert float((out.cpu() - t(z['out'])).abs().max()) <= step * 1.001


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_ste_backward_large_and_layouts(q, dtype):
    """Vectorised STE backward at a BERT-sized tensor, plus per-embedding params and a ragged,
    unaligned tensor through the scalar kernel; dx against autograd on the CPU oracle."""
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(64, 128, 768, generator=g) * 3).to(dtype)
    go = torch.randn(64, 128, 768, generator=g).to(dtype)
    delta, zf = O.asym_params_from_range(-4.0, 5.0, 8)
    gx, gd, gz = be.fake_quant_bwd(x.to(DEV), go.to(DEV), delta.to(DEV), zf.to(DEV), None, 8, False, False,
                                   1e-8, 1, 1, param_grads=True)
    _, ref_dx, ref_dd, ref_dz = O.fake_quant_with_grads(x.float(), delta, zf, 8, False, grad_out=go.float())
    assert torch.equal(gx.cpu(), ref_dx.to(dtype))
    assert torch.allclose(gd.cpu().reshape(()), ref_dd, rtol=2e-4, atol=1e-2)
    assert torch.allclose(gz.cpu().reshape(()), ref_dz, rtol=2e-4, atol=1e-2)
    # per-embedding parameters (scalar kernel), ragged + unaligned view
    xs, gs = x[:3, :5].contiguous(), go[:3, :5].contiguous()
    dv, zv = O.asym_params_from_range(xs.float().amin((0, 1)), xs.float().amax((0, 1)), 4)
    gx2, _, _ = be.fake_quant_bwd(xs.to(DEV), gs.to(DEV), dv.to(DEV), zv.to(DEV), None, 4, False, False, 1e-8,
                                  768, 1)
    _, ref2, _, _ = O.fake_quant_with_grads(xs.float(), dv, zv, 4, False, grad_out=gs.float(), axis=2)
    assert torch.equal(gx2.cpu(), ref2.to(dtype))
    flat = x.reshape(-1)[1:4098].to(DEV)
    gflat = go.reshape(-1)[1:4098].to(DEV)
    gx3, _, _ = be.fake_quant_bwd(flat, gflat, delta.to(DEV), zf.to(DEV), None, 8, False, False, 1e-8, 1, 1)
    assert torch.equal(gx3.cpu(), ref_dx.reshape(-1)[1:4098].to(dtype))


def test_percentile_ranges_on_gpu(q):
    g = torch.Generator().manual_seed(12)
    w = torch.randn(768, 3072, generator=g) * 0.05
    for p in (0.01, 1.0):
        est = q.RangeEstimators.current_minmax.cls(percentile=p, per_channel=True)
        lo, hi = est(w.to(DEV))
        r_lo, r_hi = np.percentile(w.numpy(), (p, 100 - p), axis=-1)
        assert torch.equal(lo.cpu(), torch.Tensor(r_lo)) and torch.equal(hi.cpu(), torch.Tensor(r_hi))
        mgr = q.QuantizationManager(qmethod=q.QMethods.symmetric_uniform, init=q.RangeEstimators.current_minmax,
                                    per_channel=True, qparams=dict(n_bits=4), init_params=dict(percentile=p))
        y = mgr(w.to(DEV))
        d, s = O.sym_params_from_range(torch.Tensor(r_lo), torch.Tensor(r_hi), 4)
        _, ref = O.fake_quant(w, d, None, 4, True, bool(s), per_channel=True)
        assert torch.equal(y.cpu(), ref)


def test_order_statistics_radix_select_equals_a_full_sort():
    """tq_order_stats (csrc/tq_select.hip): the elements of given ranks of every row, selected by an MSB-first radix
    select -- against torch.sort on the CPU, exactly (bit patterns), on: one long row (per-tensor activation statistics,
    grid-wide histogram passes), many short rows (per-channel, one block per row), heavy ties, infinities, NaN (sorts
    last), denormals, a single-element row, bf16 storage, ranks 0 and n - 1."""
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(31)

    def check(x, ranks):
        got = be.order_stats(x.to(DEV), ranks).cpu()
        srt, _ = torch.sort(x.float(), dim=-1)
        want = torch.stack([srt[:, r] for r in ranks], dim=1)
        # (bit-exact except the sign of a zero inside a run of +-0 ties, which a comparison sort leaves unspecified)
        same = (got.view(torch.int32) == want.view(torch.int32)) | ((got == 0) & (want == 0)) | (got.isnan() & want.isnan())
        assert bool(same.all()), (x.shape, ranks, got[~same][:4], want[~same][:4])

    x = torch.randn(1, 786432, generator=g)                          # [8, 128, 768] activations as one row
    x[0, ::97] *= 40.0
    n = x.shape[1]
    check(x, [0, 1, n // 2, n - 1])
    check(x, [78, 79, n - 79, n - 78])                               # the ranks of percentile 0.01 and 99.99
    check(x.to(torch.bfloat16), [5, n - 6])
    w = torch.randn(3072, 768, generator=g) * 0.05                   # per-channel weights: one block per row
    check(w, [0, 7, 760, 767])
    check(w[:40], [383, 384])                                        # few short rows
    t = torch.randint(-3, 4, (4, 20000), generator=g).float()        # heavy ties (7 distinct values)
    check(t, [0, 9999, 10000, 19999])
    s = torch.randn(2, 70000, generator=g)
    s[0, :5] = float('inf')
    s[0, 5:9] = float('-inf')
    s[1, :3] = float('nan')
    s[1, 3:6] = 1e-42                                                # denormals
    s[1, 6] = -0.0
    check(s, [0, 3, 69996, 69999])
    check(torch.randn(100, 10000, generator=g), [0, 4999, 5000, 9999])         # many rows AND long rows: one block per row
    check(torch.randn(64, 5000, generator=g).to(torch.float16), [17, 4000])     # the last row count of the grid-wide mode, fp16
    check(torch.tensor([[2.5]]), [0, 0])
    with pytest.raises(_hip.TQError, match='rank'):
        be.order_stats(w.to(DEV), [768])
    with pytest.raises(_hip.TQError, match='ranks per call'):
        be.order_stats(w.to(DEV), [0, 1, 2, 3, 4])


def test_percentile_per_tensor_on_gpu_equals_numpy(q):
    """CurrentMinMaxEstimator(percentile=p) on an activation tensor (reference range_estimators.py:131-140: to_numpy +
    np.percentile(data, (p, 100)), quirk q6: the upper end is the maximum) -- now without a device sort and without the
    host round trip of the tensor: bit-equal to numpy."""
    g = torch.Generator().manual_seed(13)
    x = torch.randn(8, 128, 768, generator=g)
    x[..., 308] *= 20
    for p in (0.01, 0.5, 1.0, 25.0):
        est = q.RangeEstimators.current_minmax.cls(percentile=p)
        lo, hi = est(x.to(DEV))
        r = np.percentile(x.numpy(), (p, 100))
        assert torch.equal(lo.cpu().reshape(-1), torch.Tensor(np.atleast_1d(r[0]))), p
        assert torch.equal(hi.cpu().reshape(-1), torch.Tensor(np.atleast_1d(r[1]))), p


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('rows', [400, 1100, 2600], ids=['400-partials', '1100-partials', '2600-partials'])
def test_fused_calibration_step_at_sizes_across_its_launch_forms(rows, dtype):
    """tq_calibrate_tensor picks its launch form by the number of block partials of the statistics pass: the ticket kernel /
    two ticket-free launches up to 512, two ticket-free launches up to 2048 (round 6: sites of [128,128]-token calibrating
    forwards), four launches beyond.  Three running-min/max batches (fresh and in-place state) against the separate calls
    tq_minmax -> tq_range_update -> tq_set_range_asym -> tq_fake_quant_fwd: state, parameters and y bit-equal."""
    from quantization import _hip
    be = _hip.backend()
    per_block = 256 * (4 if dtype == torch.float32 else 8) * 8
    n = rows * per_block - 3 * (4 if dtype == torch.float32 else 8)
    g = torch.Generator().manual_seed(rows)
    xs = [(torch.randn(n, generator=g) * (1.0 + 0.5 * i)).to(dtype).to(DEV) for i in range(3)]
    xs[2][n // 2] = 40.0
    lo = hi = None
    state = None
    inplace = None
    for i, x in enumerate(xs):
        mn, mx = be.minmax(x, 1, 1)
        if lo is None:
            lo, hi = mn.clone(), mx.clone()
        else:
            lo, hi = be.range_update(_hip.EST_RUNNING, mn, mx, lo, hi, 0.9)
        d, z = be.set_range_asym(lo, hi, 8, 1e-8, False)
        y_ref, _ = be.fake_quant(x, d, z, None, 8, False, False, 1e-8, 1, 1)
        r = be.calibrate_minmax(x, 1, 1, _hip.EST_RUNNING, None if state is None else state[0], None if state is None else state[1],
                                0.9, 0, None, 8, False, 1e-8, False)
        state = (r[0], r[1])
        if inplace is None:
            inplace = tuple(None if v is None else v.clone() for v in r[:5])
            ri = r
        else:
            ri = be.calibrate_minmax(x, 1, 1, _hip.EST_RUNNING, inplace[0], inplace[1], 0.9, 0, None, 8, False, 1e-8, False, out=inplace)
        for got in (r, ri):
            assert torch.equal(got[0].reshape(()), lo.reshape(())) and torch.equal(got[1].reshape(()), hi.reshape(())), (rows, i)
            assert torch.equal(got[2].reshape(()), d.reshape(())) and torch.equal(got[3].reshape(()), z.reshape(())), (rows, i)
            assert torch.equal(got[5].view(torch.int16 if got[5].element_size() == 2 else torch.int32),
                               y_ref.view(torch.int16 if y_ref.element_size() == 2 else torch.int32)), (rows, i)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('symmetric', [False, True], ids=['asym', 'sym'])
@pytest.mark.parametrize('mode', [0, 1, 2], ids=['current', 'all', 'running'])
def test_split_calibration_step_equals_the_single_gpu_step(mode, symmetric, dtype):
    """tq_calibrate_stats + tq_calibrate_apply (the sharded step: statistics | exchange | update + quantize) against
    tq_calibrate_minmax / tq_calibrate_tensor (one GPU) on the same tensor, over three batches: estimator state,
    parameters and y bit-equal.  Fresh output buffers take the ONE-launch apply (fq_tensor_calib: every block re-derives
    the parameters from the statistics), in-place state the separate update launch: both are checked, on a vectorised
    tensor, a ragged one (scalar tail) and an unaligned view."""
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(5 + mode)
    base = [torch.randn(3 * 128 * 96 + 8, generator=g) * (1.0 + i) for i in range(3)]
    # degenerate batches: a constant tensor (zero-width range -> eps), +-inf entries, a NaN (poisons the state)
    const = torch.full_like(base[0], 0.75)
    infs = base[1].clone()
    infs[17], infs[4001] = float('inf'), float('-inf')
    nans = base[2].clone()
    nans[123] = float('nan')
    base = base + [const, infs, nans]
    for view in ('vec', 'ragged', 'unaligned'):
        def cut(t):
            t = t.to(dtype).to(DEV)
            if view == 'vec':
                return t[:3 * 128 * 96].reshape(3, 128, 96)
            if view == 'ragged':
                return t[:3 * 128 * 96 + 5]
            return t[1:3 * 128 * 96 + 1]
        ref = [None, None]
        fresh = [None, None]
        inplace = None
        for i, t in enumerate(base):
            x = cut(t)
            r = be.calibrate_minmax(x, 1, 1, mode, ref[0], ref[1], 0.9, 0, None, 8, symmetric, 1e-8, False)
            ref = [r[0], r[1]]
            f = be.calibrate_apply(be.calibrate_stats(x, 1, 1), x, 1, 1, mode, fresh[0], fresh[1], 0.9, 0, None, 8,
                                   symmetric, 1e-8, False)
            fresh = [f[0], f[1]]
            if inplace is None:
                inplace = tuple(None if v is None else v.clone() for v in f[:5])
                ip = f
            else:
                ip = be.calibrate_apply(be.calibrate_stats(x, 1, 1), x, 1, 1, mode, inplace[0], inplace[1], 0.9, 0, None, 8,
                                        symmetric, 1e-8, False, out=inplace)
            for a, b, c in zip(r, f, ip):
                if a is None:
                    assert b is None and c is None
                else:
                    # (NaN == NaN here: a poisoned state must be poisoned identically on both paths)
                    assert torch.equal(a.float().nan_to_num(nan=-7.0), b.float().nan_to_num(nan=-7.0)), (view, i)
                    assert torch.equal(a.float().nan_to_num(nan=-7.0), c.float().nan_to_num(nan=-7.0)), (view, i)
