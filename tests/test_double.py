"""`--double` (reference main.py:227-231): the quantizer path in float64.

GPU: every f64 kernel against the oracle (dtype-generic torch restatement) on float64 inputs -- bit-exact for the
element-wise work (IEEE double division / rint / clamp / mul on both sides), exact for min / max and the range ->
parameter formulas, 1e-13 relative for reductions whose summation order is not pinned (parameter gradients, MSE
losses).  Fixtures: tests/golden/double.npz, written by the reference's own quantizers / estimators on float64
tensors, and tests/golden/bert_2l_double.npz, a 2-layer BERT-base-width model driven through the reference's
quantized blocks after `m.double()` (tests/golden/make_golden.py `gen_double`, make_golden_bert.py `double`)."""
import os

import numpy as np
import pytest
import torch

from oracle import tq_oracle as O
from tests.conftest import GOLDEN


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


LAYOUTS = [((64, 96), None, False), ((64, 96), None, True), ((4, 33, 48), 2, False), ((4, 33, 48), 1, False),
           ((7, 5), None, False), ((3, 1027), None, True)]


@pytest.mark.gpu
@pytest.mark.parametrize('shape,axis,per_channel', LAYOUTS)
@pytest.mark.parametrize('symmetric', [False, True])
@pytest.mark.parametrize('n_bits', [4, 8, 16])
def test_fake_quant_f64_matches_oracle(shape, axis, per_channel, symmetric, n_bits):
    from quantization.quantizers import AsymmetricUniformQuantizer, SymmetricUniformQuantizer
    if symmetric and axis is not None:
        pytest.skip('symmetric + axis raises upstream too (quirk q3)')
    x = _rand(shape, 11 + n_bits, 2.0)
    x.view(-1)[:4] = torch.tensor([0.0, -0.0, 1e300, -1e300], dtype=torch.float64)
    if axis is not None:
        mn, mx = O.minmax_axis(x, axis)
    elif per_channel:
        mn, mx = O.minmax_channel(x)
    else:
        mn, mx = O.minmax_tensor(x)
    mn, mx = mn * 0.7, mx * 0.7                                  # clip something
    cls = SymmetricUniformQuantizer if symmetric else AsymmetricUniformQuantizer
    q = cls(n_bits, per_channel=per_channel, axis=axis).cuda()
    q.set_quant_range(mn.cuda(), mx.cuda())
    if symmetric:
        d_ref, s_ref = O.sym_params_from_range(mn, mx, n_bits)
        assert q._delta.dtype == torch.float64 and torch.equal(q._delta.cpu().view(-1), d_ref.view(-1))
        assert bool(q._signed) == bool(s_ref)
        zf = None
    else:
        d_ref, zf = O.asym_params_from_range(mn, mx, n_bits)
        assert q._delta.dtype == torch.float64 and torch.equal(q._delta.cpu().view(-1), d_ref.view(-1))
        assert torch.equal(q._zero_float.cpu().view(-1), zf.view(-1))
    y = q(x.cuda())
    xi = q.to_integer_forward(x.cuda())
    ref_i, ref_y = O.fake_quant(x, d_ref, zf, n_bits, symmetric, signed=bool(s_ref) if symmetric else False,
                                axis=axis, per_channel=per_channel)
    assert y.dtype == torch.float64 and xi.dtype == torch.float64
    assert torch.equal(xi.cpu(), ref_i)
    assert torch.equal(y.cpu(), ref_y)


@pytest.mark.gpu
def test_fp32_range_buffers_are_widened_exactly():
    """A range set from python floats is an fp32 buffer (quantizers.py:248-250); float64 data divided by it is a
    float64 operation on the widened value -- torch's type promotion, reproduced by the host side."""
    from quantization.quantizers import AsymmetricUniformQuantizer
    x = _rand((5, 77), 3, 3.0)
    q = AsymmetricUniformQuantizer(8).cuda()
    q.set_quant_range(-1.7, 2.9)
    assert q._delta.dtype == torch.float32
    d, zf = O.asym_params_from_range(-1.7, 2.9, 8)
    _, ref = O.fake_quant(x, d, zf, 8, False)
    assert ref.dtype == torch.float64
    assert torch.equal(q(x.cuda()).cpu(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize('shape,n_params,inner', [((8, 128, 96), 1, 1), ((64, 300), 64, 300), ((6, 50, 96), 96, 1),
                                                  ((6, 50, 24), 24, 1), ((5, 7, 9), 7, 9), ((1, 1), 1, 1)])
def test_minmax_f64(shape, n_params, inner):
    from quantization import _hip
    x = _rand(shape, 5)
    mn, mx = _hip.backend().minmax(x.cuda(), n_params, inner)
    v = x.reshape(-1, n_params, inner)
    assert mn.dtype == torch.float64
    assert torch.equal(mn.cpu().view(-1), v.amin(dim=(0, 2))) and torch.equal(mx.cpu().view(-1), v.amax(dim=(0, 2)))
    k = min(3, x.numel() - 1)
    x.view(-1)[k] = float('nan')
    mn, mx = _hip.backend().minmax(x.cuda(), n_params, inner)
    assert torch.isnan(mn.view(-1)[0 if n_params == 1 else (k // inner) % n_params])


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['current', 'all', 'running'])
@pytest.mark.parametrize('n_groups', [0, 4])
def test_estimators_f64_match_oracle(mode, n_groups):
    from quantization.range_estimators import (AllMinMaxEstimator, CurrentMinMaxEstimator, RunningMinMaxEstimator)
    cls = {'current': CurrentMinMaxEstimator, 'all': AllMinMaxEstimator, 'running': RunningMinMaxEstimator}[mode]
    if mode == 'all' and n_groups:
        pytest.skip('AllMinMax ignores axis upstream')
    kw = dict(axis=2, n_groups=n_groups) if n_groups else {}
    est = cls(**kw)
    cur = None
    for b in range(3):
        x = _rand((3, 17, 32), 40 + b, 1.0 + b)
        got = est(x.cuda())
        if n_groups:
            mn, mx = O.minmax_groups(x, 2, n_groups)
        else:
            mn, mx = O.minmax_tensor(x)
        if mode == 'current' or cur is None:
            cur = (mn, mx)
        elif mode == 'all':
            cur = O.allminmax_update(cur[0], cur[1], mn, mx)
        else:
            cur = O.running_update(cur[0], cur[1], mn, mx, 0.9)
        assert got[0].dtype == torch.float64
        assert torch.equal(got[0].cpu().view(-1), cur[0].view(-1)) and torch.equal(got[1].cpu().view(-1), cur[1].view(-1))


@pytest.mark.gpu
@pytest.mark.parametrize('shape,axis,per_channel', [((64, 96), None, False), ((16, 40), None, True), ((3, 20, 24), 2, False)])
def test_ste_backward_f64(shape, axis, per_channel):
    from quantization.quantizers import AsymmetricUniformQuantizer
    x = _rand(shape, 9, 2.0)
    g = _rand(shape, 10)
    if axis is not None:
        mn, mx = O.minmax_axis(x, axis)
    elif per_channel:
        mn, mx = O.minmax_channel(x)
    else:
        mn, mx = O.minmax_tensor(x)
    q = AsymmetricUniformQuantizer(6, per_channel=per_channel, axis=axis).cuda()
    q.set_quant_range((mn * 0.6).cuda(), (mx * 0.6).cuda())
    q(x.cuda())                        # brings vector ranges into broadcast layout (as a calibration pass does)
    q.make_range_trainable()
    xg = x.cuda().requires_grad_(True)
    y = q(xg)
    y.backward(g.cuda())
    # oracle: autograd through the reference's expression
    d, zf = q._delta.detach().cpu(), q._zero_float.detach().cpu()
    if per_channel:
        d, zf = d.view(-1, 1), zf.view(-1, 1)
    _, dx, dd, dz = O.fake_quant_with_grads(x, d, zf, 6, False, grad_out=g, axis=axis)
    assert torch.equal(xg.grad.cpu(), dx)
    for got, want in ((q._delta.grad, dd), (q._zero_float.grad, dz)):
        got, want = got.cpu().view(-1), want.view(-1)
        assert torch.allclose(got, want, rtol=1e-12, atol=1e-12 * float(want.abs().max() + 1e-300))


@pytest.mark.gpu
def test_mse_estimator_f64_grid_and_golden_section():
    from quantization.quantizers import AsymmetricUniformQuantizer, SymmetricUniformQuantizer
    from quantization.range_estimators import MSE_Estimator, OptMethod
    x = _rand((32, 200), 21, 1.3)
    for qcls, opt in ((SymmetricUniformQuantizer, OptMethod.grid), (AsymmetricUniformQuantizer, OptMethod.grid),
                      (SymmetricUniformQuantizer, OptMethod.golden_section)):
        q = qcls(4).cuda()
        est = MSE_Estimator(quantizer=q, opt_method=opt, num_candidates=20)
        xmin, xmax = est(x.cuda())
        # the losses the search saw, against the oracle's float64 evaluation of the same candidates
        spec = O.QSpec(4, qcls is SymmetricUniformQuantizer)
        loss = est._loss_dev.cpu().numpy().reshape(-1)
        if opt == OptMethod.grid:
            thr = est._thr_dev.cpu().numpy()
            for c in (0, len(loss) // 2, len(loss) - 1):
                want = float(O.mse_loss_value(spec, x, float(thr[0, c]), float(thr[1, c])))
                assert abs(loss[c] - want) <= 1e-12 * want, (c, loss[c], want)
            best = int(np.argmin(loss))
            assert float(xmax) == float(thr[1, best])
        else:
            assert 0.0 < float(xmax) < float(x.abs().max()) + 0.5


def _fx(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip(f'{name} not generated')
    return np.load(path, allow_pickle=False)


def _run_double_fixture(device):
    """Outputs of the REFERENCE's quantizers / estimators on float64 tensors (make_golden.py gen_double)."""
    from quantization.quantization_manager import QuantizationManager
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    z = _fx('double.npz')
    for i in range(int(z['n_cases'])):
        method = QMethods[str(z[f'c{i}_method'])]
        est = RangeEstimators[str(z[f'c{i}_estimator'])]
        per_channel, axis = bool(z[f'c{i}_per_channel']), int(z[f'c{i}_axis'])
        qparams = dict(n_bits=int(z[f'c{i}_n_bits']))
        init_params = dict(num_candidates=20) if est == RangeEstimators.MSE else {}
        m = QuantizationManager(method, init=est, per_channel=per_channel, qparams=qparams, init_params=init_params,
                                axis=axis if axis >= 0 else None).to(device)
        m.estimate_ranges()
        xs = z[f'c{i}_x']
        for b in range(xs.shape[0]):
            y = m(torch.from_numpy(xs[b]).to(device))
        assert y.dtype == torch.float64
        d = m.quantizer._delta.detach().cpu().numpy().reshape(-1)
        assert d.dtype == z[f'c{i}_delta'].dtype, (i, d.dtype)         # float32 after MSE (python-float thresholds)
        assert np.array_equal(d, z[f'c{i}_delta'].reshape(-1)), i
        if f'c{i}_zero_float' in z.files:
            assert np.array_equal(m.quantizer._zero_float.detach().cpu().numpy().reshape(-1),
                                  z[f'c{i}_zero_float'].reshape(-1)), i
        assert np.array_equal(y.cpu().numpy(), z[f'c{i}_y']), i
        m.fix_ranges()
        assert np.array_equal(m(torch.from_numpy(xs[0]).to(device)).cpu().numpy(), z[f'c{i}_y_fixed']), i


@pytest.mark.gpu
def test_reference_double_fixture_gpu():
    _run_double_fixture('cuda')


def test_reference_double_fixture_host_logic_cpu():
    """The same fixture through the host classes over the oracle-backed backend double (no GPU)."""
    from quantization import _hip
    from tests._oracle_backend import OracleBackend
    prev = _hip.set_backend(OracleBackend())
    try:
        _run_double_fixture('cpu')
    finally:
        _hip.set_backend(prev)


# ---- whole model: 2-layer BERT-base-width, W8A8, `m.double()` (make_golden_bert.py double) -------------------------
def _double_bert(device):
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from tests.harness_bert import build_bert_base
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
              weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)
    model, hf = build_bert_base(seed=1000, num_layers=2, **qp)
    model = model.to(device).eval()
    for m in model.modules():                       # reference main.py:227-231
        if hasattr(m, 'weight') or hasattr(m, 'bias'):
            m.double()
    return model, hf


def _run_double_bert(device):
    from tests.harness_bert import quantizer_census
    from utils.utils import pass_data_for_range_estimation
    z = _fx('bert_2l_double.npz')
    model, hf = _double_bert(device)
    from tests.conftest import check_weights_reproduced
    check_weights_reproduced(hf, z)              # build-independent numpy-stream weights: an assertion, not a skip
    ids = torch.from_numpy(z['input_ids'])[:4, :64]
    with torch.no_grad():
        pass_data_for_range_estimation([(ids,)], model, act_quant=True, weight_quant=True, max_num_batches=1)
        model.fix_ranges()
        logits = model(ids.to(device))
    act, wts = quantizer_census(model)
    act = [(n, m) for n, m in act if m.quantizer.is_initialized]
    wts = [(n, m) for n, m in wts if m.quantizer.is_initialized]
    assert len(act) == len(z['act_min']) and len(wts) == len(z['w_delta'])
    assert logits.dtype == torch.float64
    assert all(m.quantizer._delta.dtype == torch.float64 for _, m in act + wts)
    wd = np.array([float(m.quantizer._delta) for _, m in wts])
    amin = np.array([float(m.range_estimator.current_xmin) for _, m in act])
    amax = np.array([float(m.range_estimator.current_xmax) for _, m in act])
    return z, wd, amin, amax, logits.cpu().numpy()


def test_double_bert_cpu_exact():
    """Host classes over the oracle-backed double: every float64 range, weight delta and logit of the reference."""
    from quantization import _hip
    from tests._oracle_backend import OracleBackend
    prev = _hip.set_backend(OracleBackend())
    try:
        torch.set_num_threads(8)
        z, wd, amin, amax, logits = _run_double_bert('cpu')
        assert np.array_equal(wd, z['w_delta'])
        assert np.array_equal(amin, z['act_min']) and np.array_equal(amax, z['act_max'])
        assert np.array_equal(logits, z['logits'])
    finally:
        _hip.set_backend(prev)
        torch.set_num_threads(1)


@pytest.mark.gpu
def test_double_bert_gpu():
    """Through the f64 kernels.  Weight deltas and the pre-GEMM sites are exact; behind float64 GEMMs (rocBLAS vs
    the CPU's) statistics agree to ~1e-13 relative unless an element sits within that distance of a rounding
    boundary -- a flipped index moves a downstream extreme by a visible amount, so the bars are the float64 analogue of
    test_bert_e2e's: exact where no GEMM is upstream, tight quantiles elsewhere."""
    z, wd, amin, amax, logits = _run_double_bert('cuda')
    assert np.array_equal(wd, z['w_delta'])
    span = z['act_max'] - z['act_min']
    rel = np.maximum(np.abs(amin - z['act_min']), np.abs(amax - z['act_max'])) / span
    assert rel[0] == 0 and rel[1] == 0
    assert np.median(rel) <= 1e-9 and rel.max() <= 2e-2, (np.median(rel), rel.max())
    lspan = float(z['logits'].max() - z['logits'].min())
    assert np.abs(logits - z['logits']).max() <= 2e-2 * lspan
