"""K independent golden-section range searches in lock step (range_estimators.golden_section_lockstep,
autoquant_utils.precalibrate_weights): the thresholds are those of the sequential searches bit for bit -- same scipy
calls on the same fp32 loss values (reference range_estimators.py:296-327, 422-470; README.md:149-157) -- with one
device->host copy per ROUND instead of per evaluation.  CPU: through the oracle backend; GPU: through the kernels, plus
the wall time of the README recipe's 102 weight searches before / after.
"""
import time

import numpy as np
import pytest
import torch


def _estimators(symmetric, n_bits, n):
    from quantization.quantizers import QMethods
    from quantization.range_estimators import OptMethod, RangeEstimators
    qcls = (QMethods.symmetric_uniform if symmetric else QMethods.asymmetric_uniform).cls
    return [RangeEstimators.MSE.cls(quantizer=qcls(n_bits=n_bits), opt_method=OptMethod.golden_section) for _ in range(n)]


def _tensors(device, one_sided=False):
    g = torch.Generator().manual_seed(11)
    shapes = [(64, 96), (768,), (30, 40, 8), (3072 // 8, 96), (5, 7)]
    xs = [torch.randn(*s, generator=g) * (0.02 * (i + 1)) for i, s in enumerate(shapes)]
    if one_sided:
        xs = [x.abs() for x in xs]
    return [x.to(device) for x in xs]


def _check_lockstep_equals_sequential(device):
    from quantization.range_estimators import golden_section_lockstep
    for symmetric, n_bits, one_sided in ((True, 8, False), (True, 4, False), (False, 4, False), (False, 8, True)):
        xs = _tensors(device, one_sided)
        seq = _estimators(symmetric, n_bits, len(xs))
        ref = [e(x) for e, x in zip(seq, xs)]
        par = _estimators(symmetric, n_bits, len(xs))
        stats = golden_section_lockstep(list(zip(par, xs)))
        assert stats['searches'] == len(xs) and stats['evaluations'] > stats['rounds'] >= 10, stats
        # nested searches (asymmetric two-sided) run ~20 x 20 evaluations per tensor; rounds = the longest search
        assert stats['rounds'] * len(xs) >= stats['evaluations'] > 2 * stats['rounds'], stats
        for e, p, (rmin, rmax) in zip(seq, par, ref):
            assert torch.equal(p.current_xmin.cpu(), rmin.cpu()) and torch.equal(p.current_xmax.cpu(), rmax.cpu())
            assert p.result.x == e.result.x and p.result.nfev == e.result.nfev
            assert p.max_search_range == e.max_search_range and p.one_sided_dist == e.one_sided_dist
        # the estimating forward that follows answers from the memo: no launch, same tensors back
        from quantization import _hip
        be = _hip.backend()
        calls = []
        orig = be.mse_candidates_ordered
        be.mse_candidates_ordered = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            for p, x, (rmin, rmax) in zip(par, xs, ref):
                mn, mx = p(x)
                assert torch.equal(mn.cpu(), rmin.cpu()) and torch.equal(mx.cpu(), rmax.cpu())
            assert not calls
            # ... and only for THAT tensor in THAT state: an in-place change, or another tensor, searches again
            xs[0].mul_(1.5)
            par[0](xs[0])
            assert calls
            n = len(calls)
            par[1](xs[1].clone())
            assert len(calls) > n
            par[2].reset()
            assert par[2]._memo is None
        finally:
            be.__dict__.pop('mse_candidates_ordered', None)


def test_lockstep_equals_sequential_cpu_oracle():
    from quantization import _hip
    from tests._oracle_backend import OracleBackend
    prev = _hip.set_backend(OracleBackend())
    try:
        _check_lockstep_equals_sequential('cpu')
    finally:
        _hip.set_backend(prev)


def test_restated_bounded_brent_is_scipys():
    """`range_estimators._bounded_brent` (scipy's `minimize_scalar(method='Bounded')` as a resumable generator) against
    scipy itself: abscissa, value, evaluation count AND the numpy scalar types of the result on 2 000 random fp32-valued
    objectives -- smooth, kinked, noisy, with plateaus (quantization losses are piecewise smooth), flat, and with NaN."""
    from scipy.optimize import minimize_scalar
    from quantization.range_estimators import _bounded_brent
    rs = np.random.RandomState(0)

    def run(f, bounds):
        g = _bounded_brent(bounds)
        x = next(g)
        try:
            while True:
                x = g.send(f(x))
        except StopIteration as stop:
            return stop.value

    for t in range(2000):
        c, k, n, noise = rs.uniform(0.05, 3), rs.uniform(0.1, 5), rs.randint(2, 6), rs.uniform(0, 1e-3)
        lo = rs.uniform(0.001, 0.05)
        hi = lo + rs.uniform(0.5, 4)
        kind = t % 5

        def f(x):
            if kind == 3:
                return np.float32(1.25)                                    # flat
            if kind == 4 and x > 0.5 * (lo + hi):
                return np.float32('nan')
            return np.float32(k * abs(x - c) ** (n / 2.0) + noise * np.sin(1000 * x) + (0.3 * np.floor(8 * x) / 8 if kind == 2 else 0))
        ref = minimize_scalar(f, bounds=(lo, hi), method='Bounded')
        got = run(f, (lo, hi))
        assert type(got.x) is type(ref.x) and type(got.fun) is type(ref.fun), t
        same_x = got.x == ref.x or (np.isnan(got.x) and np.isnan(ref.x))
        same_f = got.fun == ref.fun or (np.isnan(got.fun) and np.isnan(ref.fun))
        assert same_x and same_f and got.nfev == ref.nfev and got.status == ref.status and got.success == ref.success, (t, got, ref)
    # float64 objective values (--double)
    for t in range(200):
        c = rs.uniform(0.1, 2)
        f = lambda x: np.float64((x - c) ** 2 + 0.01 * np.abs(np.sin(50 * x)))
        ref = minimize_scalar(f, bounds=(0.01, 3.0), method='Bounded')
        got = run(f, (0.01, 3.0))
        assert got.x == ref.x and got.fun == ref.fun and got.nfev == ref.nfev


def test_lockstep_propagates_a_failing_evaluation():
    from quantization import _hip
    from quantization.range_estimators import golden_section_lockstep
    from tests._oracle_backend import OracleBackend
    be = OracleBackend()
    prev = _hip.set_backend(be)
    try:
        xs = _tensors('cpu')
        est = _estimators(True, 8, len(xs))
        orig = be.mse_candidates_ordered
        bad = xs[2]

        def flaky(x, *a, **k):
            if x.shape == bad.shape:
                raise RuntimeError('injected failure')
            return orig(x, *a, **k)
        be.mse_candidates_ordered = flaky
        with pytest.raises(RuntimeError, match='injected failure'):
            golden_section_lockstep(list(zip(est, xs)))
        assert all(e._memo is None for e in est)
    finally:
        _hip.set_backend(prev)


def _toy(device, golden=True):
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_model import QuantizedModel
    from quantization.quantizers import QMethods
    from quantization.range_estimators import OptMethod, RangeEstimators
    torch.manual_seed(5)

    class Net(QuantizedModel):
        def __init__(self):
            super().__init__()
            qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
                      weight_range_method=RangeEstimators.MSE if golden else RangeEstimators.current_minmax,
                      weight_range_options=dict(opt_method=OptMethod.golden_section) if golden else {},
                      act_range_method=RangeEstimators.current_minmax)
            self.a = quantize_model(torch.nn.Linear(32, 64), **qp)
            self.ln = quantize_model(torch.nn.LayerNorm(64), **qp)
            self.b = quantize_model(torch.nn.Linear(64, 32), **qp)

        def forward(self, x):
            return self.b(self.ln(self.a(x)))
    return Net().to(device).eval()


def _check_precalibrate(device):
    from quantization.autoquant_utils import precalibrate_weights
    from quantization.quantization_manager import QuantizationManager
    from utils.utils import pass_data_for_range_estimation
    x = torch.randn(4, 6, 32, generator=torch.Generator().manual_seed(2)).to(device)

    def deltas(model):
        return [m.quantizer._delta.detach().cpu().clone() for n, m in model.named_modules()
                if isinstance(m, QuantizationManager) and n.endswith('weight_quantizer')]

    # (a) the calibration driver runs the lock-step search by itself (options.LOCKSTEP_WEIGHT_SEARCH, on by default) ...
    from quantization import options
    assert options.LOCKSTEP_WEIGHT_SEARCH is True
    m1 = _toy(device)
    with torch.no_grad():
        pass_data_for_range_estimation([(x,)], m1, act_quant=True, weight_quant=True, max_num_batches=1)
    assert all(m.range_estimator._memo is not None for n, m in m1.named_modules()
               if isinstance(m, QuantizationManager) and n.endswith('weight_quantizer'))
    # (b) ... and finds what layer-by-layer searches find
    m2 = _toy(device)
    options.LOCKSTEP_WEIGHT_SEARCH = False
    try:
        with torch.no_grad():
            pass_data_for_range_estimation([(x,)], m2, act_quant=True, weight_quant=True, max_num_batches=1)
    finally:
        options.LOCKSTEP_WEIGHT_SEARCH = True
    assert all(m.range_estimator._memo is None for n, m in m2.named_modules()
               if isinstance(m, QuantizationManager) and n.endswith('weight_quantizer'))
    d1, d2 = deltas(m1), deltas(m2)
    assert len(d1) == 3 and all(torch.equal(a, b) for a, b in zip(d1, d2))
    with torch.no_grad():
        assert torch.equal(m1(x), m2(x))
    # (c) counted: three searches on a fresh model, none when nothing qualifies
    m3 = _toy(device)
    m3.set_quant_state(True, True)
    st = precalibrate_weights(m3)
    assert st['searches'] == 3 and st['rounds'] < st['evaluations']
    assert precalibrate_weights(_toy(device, golden=False))['searches'] == 0


def test_precalibrate_weights_cpu_oracle():
    from quantization import _hip
    from tests._oracle_backend import OracleBackend
    prev = _hip.set_backend(OracleBackend())
    try:
        _check_precalibrate('cpu')
    finally:
        _hip.set_backend(prev)


@pytest.mark.gpu
def test_lockstep_equals_sequential_gpu():
    _check_lockstep_equals_sequential('cuda')
    _check_precalibrate('cuda')


@pytest.mark.gpu
def test_readme_recipe_weight_calibration_wall_time():
    """The 102 weight searches of the README recipe on BERT-base: layer by layer (2 096 evaluations,
    each with its own device->host copy) against the lock-step search (26 rounds, one copy each): the SAME 102 ranges,
    wall time of both printed.  Measured: 148-156 ms layer by layer (71-75 us per evaluation) against 64.5 ms in lock step
    (2.3x; the kernels alone are ~38 ms -- a first, thread-per-search version took 250-290 ms)."""
    from harness.bert import build_bert_base
    from quantization.autoquant_utils import precalibrate_weights
    from quantization.quantizers import QMethods
    from quantization.range_estimators import OptMethod, RangeEstimators
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
              weight_range_method=RangeEstimators.MSE, weight_range_options=dict(opt_method=OptMethod.golden_section),
              act_range_method=RangeEstimators.current_minmax)
    model, _ = build_bert_base(seed=1000, **qp)
    model = model.cuda().eval()
    model.set_quant_state(True, True)
    from quantization.hijacker import QuantizationHijacker
    mods = [(mod.weight_quantizer, mod.weight) for mod in model.modules() if isinstance(mod, QuantizationHijacker)]
    assert len(mods) == 102
    torch.cuda.synchronize()
    with torch.no_grad():
        t0 = time.perf_counter()
        seq = []
        for mgr, w in mods:
            mn, mx = mgr.range_estimator(w)
            seq.append((mn.clone(), mx.clone()))
        torch.cuda.synchronize()
        t_seq = time.perf_counter() - t0
        for mgr, _ in mods:
            mgr.range_estimator.reset()
        t0 = time.perf_counter()
        st = precalibrate_weights(model)
        torch.cuda.synchronize()
        t_par = time.perf_counter() - t0
    assert st['searches'] == 102
    for (mgr, _), (mn, mx) in zip(mods, seq):
        assert torch.equal(mgr.range_estimator.current_xmin, mn) and torch.equal(mgr.range_estimator.current_xmax, mx)
    print(f'README-recipe weight calibration: layer by layer {t_seq * 1e3:.1f} ms, lock step {t_par * 1e3:.1f} ms '
          f'({t_seq / t_par:.1f}x), {st}')
    # Wall time is REPORTED, not asserted (round 6: on a busy box the first lock-step call once took 290 ms against 172 ms
    # layer by layer -- a timing flake must not fail the parity suite); the measured 2.3-2.5x is in
    # profiles/rNN/config_bench.json: config0_readme_recipe_weight_calibration.
