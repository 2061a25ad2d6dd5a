"""Host-side logic of the drop-in classes on a CPU-only box.

The HIP backend is replaced by tests/_oracle_backend.OracleBackend (the CPU oracle), so what is
under test here is everything ABOVE the C ABI: state machines, buffer names/shapes, parameter
layout inference, candidate-table construction, estimator bookkeeping, AdaRound loop, errors.
"""
import copy
import json

import numpy as np
import pytest
import torch
from torch import nn

from tests._cases import fq_case, est_inputs, t, LAYOUT_ARGS
from tests._oracle_backend import OracleBackend

torch.set_num_threads(1)


@pytest.fixture(autouse=True)
def oracle_backend():
    from quantization import _hip
    prev = _hip.set_backend(OracleBackend())
    yield
    _hip.set_backend(prev)


def _api():
    from quantization.quantizers import QMethods, QuantizerNotInitializedError
    from quantization.range_estimators import RangeEstimators, OptMethod, NoDataPassedError
    from quantization.quantization_manager import QuantizationManager, Qstates
    from utils.per_embd_quant_utils import set_act_quant_axis_and_groups
    import types
    return types.SimpleNamespace(**locals())


def _manager(q, m, init='current_minmax', init_params=None):
    la = LAYOUT_ARGS[m['layout']]
    mgr = q.QuantizationManager(qmethod=q.QMethods[m['method']], init=q.RangeEstimators[init],
                                per_channel=la['per_channel'], qparams=dict(n_bits=m['n_bits']),
                                init_params=dict(init_params or {}))
    if la['axis'] is not None:
        q.set_act_quant_axis_and_groups(mgr, axis=la['axis'], n_groups=la['n_groups'],
                                        permute=m['layout'].endswith('_perm'))
    return mgr


def test_golden_fake_quant_through_manager(golden_fake_quant):
    q = _api()
    z, meta = golden_fake_quant
    for m in meta:
        c = fq_case(z, m)
        x = c['x']
        mgr = _manager(q, m)
        if m['layout'].endswith('_perm'):
            assert mgr(x) is x
            assert torch.equal(mgr.range_estimator.ranges, c['ranges'])
            mgr.range_estimator.per_group_range_estimation = False
        y = mgr(x)
        est, qz = mgr.range_estimator, mgr.quantizer
        assert torch.equal(est.current_xmin.reshape(-1), c['xmin'].reshape(-1)), m
        assert torch.equal(est.current_xmax.reshape(-1), c['xmax'].reshape(-1)), m
        assert torch.equal(qz._delta.reshape(-1), c['delta'].reshape(-1)), m
        assert qz._delta.shape == c['delta'].shape, m           # [1,1,d] / [C,1] views like upstream
        if c['zero_float'] is not None:
            assert torch.equal(qz._zero_float.reshape(-1), c['zero_float'].reshape(-1)), m
        assert torch.equal(qz.to_integer_forward(x), c['idx']), m
        assert torch.equal(y, c['y_bf16'] if m['io'] == 'bf16' else c['y']), m
        if c['symmetric']:
            assert qz.signed == m['signed']
            assert (float(qz.int_min), float(qz.int_max)) == (m['int_min'], m['int_max'])


def test_estimator_traces(golden_estimators):
    q = _api()
    z, meta = golden_estimators
    for m in meta:
        k = m['k']
        ip = dict(m['init_params'])
        golden_section = ip.get('opt_method') == 'golden_section'
        if 'opt_method' in ip:
            ip['opt_method'] = q.OptMethod[ip['opt_method']]
        mgr = _manager(q, m, init=m['init'], init_params=ip)
        for b, x in enumerate(est_inputs(z, m)):
            y = mgr(x)
            est = mgr.range_estimator
            gmin, gmax = est.current_xmin.reshape(-1), est.current_xmax.reshape(-1)
            rmin, rmax = t(z[f'e{k}_xmin'][b]), t(z[f'e{k}_xmax'][b])
            if m['init'] == 'cross_entropy':
                assert torch.allclose(gmin, rmin, rtol=2e-3, atol=1e-4), (m, b)
                assert torch.allclose(gmax, rmax, rtol=2e-3, atol=1e-4), (m, b)
            else:
                assert torch.equal(gmin, rmin), (m, b)
                assert torch.equal(gmax, rmax), (m, b)
        if m['init'] != 'cross_entropy':
            assert torch.equal(y, t(z[f'e{k}_y_last'])), m
        la = getattr(mgr.range_estimator, 'loss_array', None)
        if la is not None and f'e{k}_loss_array' in z.files and not golden_section:
            ref = z[f'e{k}_loss_array']
            assert la.shape == ref.shape, m
            fin = np.isfinite(ref)
            assert np.array_equal(np.isfinite(la), fin)
            if m['init'] == 'MSE':
                assert np.array_equal(la[fin], ref[fin]), m      # same fp32 sums, same fp64 accumulation
            else:
                assert np.allclose(la[fin], ref[fin], rtol=1e-5, atol=1e-7), m


def test_candidate_table_matches_reference_set_quant_range():
    """candidate_params (numpy fp32) == oracle set_quant_range + scale/zero_point properties."""
    from quantization.range_estimators import candidate_params
    from oracle import tq_oracle as O
    rng = np.random.RandomState(0)
    neg = -np.abs(rng.randn(200)) * 5
    pos = np.abs(rng.randn(200)) * 5
    neg[:20] = 0.0
    for n_bits in (2, 4, 8, 16):
        for sym in (False, True):
            tab = candidate_params(neg, pos, n_bits, sym)
            for i in range(len(neg)):
                if sym:
                    d, s = O.sym_params_from_range(float(neg[i]), float(pos[i]), n_bits)
                    lo, hi = O.grid_limits(n_bits, True, bool(s))
                    exp = [float(O.effective_scale(d)), 0.0, lo, hi]
                else:
                    d, zf = O.asym_params_from_range(float(neg[i]), float(pos[i]), n_bits)
                    exp = [float(O.effective_scale(d)), float(O.effective_zero_point(zf, n_bits)),
                           0.0, 2.0 ** n_bits - 1]
                assert tab[i].tolist() == [np.float32(v) for v in exp], (n_bits, sym, i)


class ToyNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(24, 32)
        self.act = nn.GELU()
        self.ln = nn.LayerNorm(32)
        self.fc2 = nn.Linear(32, 24)


def _quant_toy(org, **qp):
    from quantization.base_quantized_model import QuantizedModel
    from quantization.base_quantized_classes import QuantizedActivation
    from quantization.autoquant_utils import quantize_model

    class QuantToy(QuantizedModel):
        def __init__(self):
            super().__init__()
            self.fc1 = quantize_model(org.fc1, **qp)
            self.act = org.act
            self.act_q = QuantizedActivation(**qp)
            self.ln = quantize_model(org.ln, **qp)
            self.fc2 = quantize_model(org.fc2, **qp)
            self.res_q = QuantizedActivation(**qp)

        def forward(self, x):
            h = self.act_q(self.act(self.fc1(x)))
            h = self.ln(h)
            return self.res_q(self.fc2(h) + x)

    return QuantToy()


def test_toy_model_calibration_state_dict(golden_toy):
    """pass_data_for_range_estimation + fix_ranges on a 2-layer QuantizedModel reproduces the
    reference's state_dict (names, shapes, values) and output."""
    q = _api()
    from utils.utils import pass_data_for_range_estimation
    z, _ = golden_toy
    org = ToyNet()
    org.load_state_dict({k[2:]: t(z[k]) for k in z.files if k.startswith('w_')})
    qp = dict(method=q.QMethods.symmetric_uniform, act_method=q.QMethods.asymmetric_uniform,
              n_bits=8, n_bits_act=8, weight_range_method=q.RangeEstimators.current_minmax,
              act_range_method=q.RangeEstimators.running_minmax)
    model = _quant_toy(org, **qp)
    loader = [(t(b),) for b in z['loader']]
    pass_data_for_range_estimation(loader, model, act_quant=True, weight_quant=True,
                                   max_num_batches=3)
    model.fix_ranges()
    model.eval()
    out = model(loader[3][0])
    sd = model.state_dict()
    names = json.loads(str(z['sd_names']))
    assert len(names) == 48
    for name in names:
        assert name in sd, name
        ref = torch.from_numpy(z['sd_' + name])
        assert sd[name].shape == ref.shape, name
        assert torch.equal(sd[name].cpu().to(ref.dtype), ref), name
    assert torch.equal(out, t(z['out']))
    # eval-mode weight cache is populated and invalidated like upstream
    assert model.fc1.cached_params is not None
    model.train()
    assert model.fc1.cached_params is None
    # state machine
    for mod in model.modules():
        if isinstance(mod, q.QuantizationManager):
            assert mod.state == q.Qstates.fix_ranges


def test_errors_and_states():
    q = _api()
    mgr = q.QuantizationManager(qmethod=q.QMethods.asymmetric_uniform, qparams=dict(n_bits=8))
    with pytest.raises(q.QuantizerNotInitializedError):
        mgr.fix_ranges()
    with pytest.raises(q.QuantizerNotInitializedError):
        _ = mgr.quantizer.delta
    with pytest.raises(q.QuantizerNotInitializedError):
        _ = q.QMethods.symmetric_uniform.cls(n_bits=8).signed
    with pytest.raises(ValueError):
        mgr.quantizer.set_quant_range(torch.zeros(3), torch.ones(3))   # vector on per-tensor
    est = q.RangeEstimators.MSE.cls(quantizer=mgr.quantizer)
    with pytest.raises(q.NoDataPassedError):
        _ = est.step_size
    with pytest.raises(NotImplementedError):
        q.RangeEstimators.MSE.cls(quantizer=None)
    x = torch.randn(4, 8)
    mgr(x)
    assert mgr.quantizer.is_initialized
    mgr.fix_ranges()
    d0 = mgr.quantizer._delta.clone()
    mgr(x * 10)
    assert torch.equal(mgr.quantizer._delta, d0)                  # fixed
    mgr.estimate_ranges_train()
    mgr.eval()
    mgr(x * 10)
    assert torch.equal(mgr.quantizer._delta, d0)                  # eval: frozen
    mgr.train()
    mgr(x * 10)
    assert not torch.equal(mgr.quantizer._delta, d0)              # train: follows the data
    mgr.reset_ranges()
    assert not mgr.quantizer.is_initialized and mgr.state == q.Qstates.estimate_ranges
    # fixed-range constructor
    m2 = q.QuantizationManager(qmethod=q.QMethods.symmetric_uniform, qparams=dict(n_bits=4),
                               x_min=-1.0, x_max=2.0)
    assert m2.state == q.Qstates.fix_ranges and m2.range_estimator is None
    assert m2.quantizer.signed is True and m2.quantizer.int_min == -8 and m2.quantizer.int_max == 7
    y = m2(x)
    assert y.shape == x.shape
    assert q.QMethods.list() == ['symmetric_uniform', 'asymmetric_uniform']
    assert q.RangeEstimators.list() == ['current_minmax', 'allminmax', 'running_minmax', 'MSE',
                                        'cross_entropy']
    # state_dict key names (checkpoint compatibility)
    keys = set(mgr.state_dict().keys()) | set(m2.state_dict().keys())
    assert 'quantizer._delta' in keys and 'quantizer._signed' in keys


def test_quant_dict_hijack():
    q = _api()
    from quantization.base_quantized_classes import QuantizedActivation, FP32Acts
    from utils.per_embd_quant_utils import hijack_act_quant
    mods = {n: QuantizedActivation(n_bits_act=8) for n in 'abcde'}
    qd = dict(a=4, b='fp32', c='per_embd', d='ng6', e='ngp3')
    for n, m in mods.items():
        hijack_act_quant(qd, n, m)
    assert mods['a'].activation_quantizer.quantizer.n_bits == 4
    assert isinstance(mods['b'].activation_quantizer, FP32Acts)
    c = mods['c'].activation_quantizer
    assert (c.axis, c.quantizer.axis, c.range_estimator.axis, c.n_groups) == (2, 2, 2, None)
    d = mods['d'].activation_quantizer
    assert d.n_groups == 6 and d.range_estimator.n_groups == 6
    assert not d.range_estimator.per_group_range_estimation
    assert mods['e'].activation_quantizer.range_estimator.per_group_range_estimation
    with pytest.raises(NotImplementedError):
        hijack_act_quant(dict(a='bogus'), 'a', QuantizedActivation())


def test_adaround_quantizer_and_fused_loop(golden_adaround):
    """AdaRoundQuantizer (alpha init, soft/hard) and the fused optimisation loop on the recorded
    batch-index sequence track the reference's autograd + torch.optim.Adam trace."""
    q = _api()
    from quantization.autoquant_utils import QuantLinear
    from quantization.adaround.quantizer import ADAROUND_QUANTIZER_MAP
    from quantization.adaround.utils import AdaRoundMode, CombinedLoss, MODE_TO_LOSS_TYPE, \
        AdaRoundTempDecayType
    from quantization.adaround.adaround import FusedAlphaAdam, optimize_local_loss
    z, meta = golden_adaround
    for m in meta:
        k = m['k']
        layer = QuantLinear(16, 12, method=q.QMethods[m['method']], n_bits=4,
                            weight_range_method=q.RangeEstimators.current_minmax)
        layer.weight.data = t(z[f'a{k}_w']).clone()
        layer.bias.data = t(z[f'a{k}_b']).clone()
        layer.quantized_weights()
        layer.caching = False
        X, tgt = t(z[f'a{k}_X']), t(z[f'a{k}_tgt'])
        with torch.no_grad():
            layer(X[:4])
        oq = layer.weight_quantizer.quantizer
        assert torch.equal(oq._delta, t(z[f'a{k}_delta']))
        wq = ADAROUND_QUANTIZER_MAP[oq.__class__](n_bits=oq.n_bits, scale_domain=oq.scale_domain,
                                                  per_channel=oq.per_channel, eps=oq.eps)
        for name in ('_delta', '_zero_float', '_signed'):
            if hasattr(oq, name):
                wq.register_buffer(name, getattr(oq, name))
        layer.weight_quantizer.quantizer = wq
        layer.weight_quantizer.fix_ranges()
        wq.round_mode = AdaRoundMode[m['mode']]
        wq.temperature = 20
        wq.soft_targets = True
        with torch.no_grad():
            soft0 = wq(layer.weight)
        assert torch.equal(wq.alpha.detach(), t(z[f'a{k}_alpha0'])), m
        assert torch.equal(soft0, t(z[f'a{k}_wq_soft0'])), m
        wq.soft_targets = False
        with torch.no_grad():
            assert torch.equal(wq(layer.weight), t(z[f'a{k}_wq_hard0'])), m
            assert torch.allclose(wq.to_integer_forward(layer.weight), t(z[f'a{k}_idx_hard0']),
                                  atol=1e-4), m
        wq.soft_targets = True
        loss_fn = CombinedLoss(quantizer=wq, loss_type=MODE_TO_LOSS_TYPE[wq.round_mode], weight=0.01,
                               max_count=m['iters'], b_range=(20, 2), warmup=0.2,
                               decay_type=AdaRoundTempDecayType.cosine, decay_shape=1.0,
                               decay_start=0.0)
        opt = FusedAlphaAdam(wq, lr=m['lr'])
        batch_idx = z[f'a{k}_batch_idx']

        class _IO:      # stands in for GetLayerInpOut: the cached I/O is (X, tgt) itself
            def __call__(self, data):
                return data, tgt[self.pos:self.pos + data.size(0)]
        io = _IO()

        def get_inp_out(data):
            pos = int((X == data[0]).all(-1).all(-1).nonzero()[0])
            return data, tgt[pos:pos + data.size(0)]

        optimize_local_loss(layer, get_inp_out, X, opt, loss_fn, m['bs'], m['iters'],
                            batch_indices=batch_idx)
        ref_alpha = t(z[f'a{k}_alphas'][-1])
        assert torch.allclose(wq.alpha.detach(), ref_alpha, rtol=1e-4, atol=1e-5), m
        wq.soft_targets = False
        with torch.no_grad():
            assert torch.equal(wq(layer.weight), t(z[f'a{k}_wq_hard1'])), m


def test_adaround_generic_autograd_path(golden_adaround):
    """An external torch optimizer over quantizer.alpha (the reference's way of driving it) goes
    through _AdaRoundFn's backward and reproduces the recorded gradients."""
    q = _api()
    from quantization.adaround.quantizer import ADAROUND_QUANTIZER_MAP
    from quantization.adaround.utils import AdaRoundMode
    z, meta = golden_adaround
    m = meta[0]
    k = m['k']
    wq = ADAROUND_QUANTIZER_MAP[q.QMethods[m['method']].cls](n_bits=4)
    wq.set_quant_range(t(z[f'a{k}_w']).min(), t(z[f'a{k}_w']).max())
    assert torch.equal(wq._delta, t(z[f'a{k}_delta']))
    wq.round_mode = AdaRoundMode[m['mode']]
    wq.temperature = 20
    wq.soft_targets = True
    w, b = t(z[f'a{k}_w']), t(z[f'a{k}_b'])
    X, tgt = t(z[f'a{k}_X']), t(z[f'a{k}_tgt'])
    idx = torch.from_numpy(z[f'a{k}_batch_idx'][0])
    out = torch.nn.functional.linear(X[idx], wq(w), b)
    loss = torch.nn.functional.mse_loss(out, tgt[idx], reduction='none').sum(1).mean()
    loss.backward()
    assert torch.allclose(wq.alpha.grad, t(z[f'a{k}_grads'][0]), rtol=1e-5, atol=1e-7)


def test_golden_section_is_chaotic_in_the_reference(golden_estimators):
    """Documents why golden-section parity is a tolerance: feeding scipy the SAME loss summed in
    fp64 instead of the reference's fp32 torch.sum moves the asymmetric 4-bit optimum to another
    local minimum of the shift loss (the optimiser's iterates are owned by scipy)."""
    from oracle import tq_oracle as O
    z, meta = golden_estimators
    m = [mm for mm in meta if mm['name'] == 'golden-asym'][0]
    x = est_inputs(z, m)[0]

    def loss64(qs, data, neg, pos, per_channel_loss=False):
        y = qs.quantize(data, x_min=neg, x_max=pos)
        return np.float32(((data - y).double() ** 2).sum().item())

    got = {}
    for name, fn in (('fp32', O.mse_loss_value), ('fp64', loss64)):
        s = O.MSESearch(O.QSpec(4, False), opt_method='golden_section', loss_value=fn)
        mn, mx = s.step_batch(x)
        got[name] = (float(mn), float(mx))
    assert got['fp32'] == (float(z[f"e{m['k']}_xmin"][0][0]), float(z[f"e{m['k']}_xmax"][0][0]))
    assert got['fp64'] != got['fp32']


def test_percentile_ranges_match_numpy():
    """percentile option of CurrentMinMaxEstimator (reference range_estimators.py:118-140): numpy is
    the arithmetic owner upstream; the device sort + float64 lerp reproduces it bit for bit,
    including the (p, 100) asymmetry of the per-tensor branch (quirk q6)."""
    q = _api()
    g = torch.Generator().manual_seed(12)
    w = torch.randn(24, 333, generator=g) * 2
    for p in (0.01, 1.0, 5.0):
        est = q.RangeEstimators.current_minmax.cls(percentile=p, per_channel=True)
        lo, hi = est(w)
        r_lo, r_hi = np.percentile(w.numpy(), (p, 100 - p), axis=-1)
        assert torch.equal(lo, torch.Tensor(r_lo)) and torch.equal(hi, torch.Tensor(r_hi))
        est = q.RangeEstimators.current_minmax.cls(percentile=p)
        lo, hi = est(w)
        r = np.percentile(w.numpy(), (p, 100))
        assert lo.shape == (1,) and float(lo) == np.float32(r[0]) and float(hi) == np.float32(r[1])


def test_adaround_samples_from_the_cached_rows():
    """Layers whose input does not carry the batch dimension (BERT's position embeddings see one [1, T]
    index tensor per forward) cache one row per BATCH, not per sample; the optimisation loop must draw its
    indices from the cached rows like the reference (adaround/adaround.py:236) -- drawing from the number
    of samples indexes past the cache (a GPU memory fault on the device path)."""
    from quantization.adaround import apply_adaround_to_layer
    from quantization.adaround.config import DEFAULT_ADAROUND_CONFIG
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_model import QuantizedModel
    from quantization.quantizers import QMethods

    class Net(QuantizedModel):
        def __init__(self):
            super().__init__()
            self.tok = quantize_model(nn.Embedding(50, 16), method=QMethods.symmetric_uniform, n_bits=4)
            self.pos = quantize_model(nn.Embedding(8, 16), method=QMethods.symmetric_uniform, n_bits=4)
            self.fc = quantize_model(nn.Linear(16, 4), method=QMethods.symmetric_uniform, n_bits=4)

        def forward(self, ids):
            pos = torch.arange(ids.shape[1]).unsqueeze(0)
            return self.fc(self.tok(ids) + self.pos(pos))

    torch.manual_seed(0)
    net = Net().eval()
    data = torch.randint(0, 50, (24, 8))
    net.set_quant_state(True, False)
    with torch.no_grad():
        net(data[:4])
    cfg = copy.deepcopy(DEFAULT_ADAROUND_CONFIG)
    cfg.iters = 12
    for name in ('pos', 'tok', 'fc'):
        net.full_precision()
        getattr(net, name).quantized_weights()
        res = apply_adaround_to_layer(net, getattr(net, name), data, batch_size=4, act_quant=False,
                                      adaround_config=copy.deepcopy(cfg))
        assert np.isfinite(res.loss_hard_after)


def test_provenance_records_die_with_any_change():
    """A provenance record (quantization/provenance.py) vouches for ONE tensor object in ONE state: views and copies
    have none, an in-place op or a re-set range of the producing quantizer invalidates it."""
    from quantization import provenance
    from quantization.quantizers import AsymmetricUniformQuantizer
    q = AsymmetricUniformQuantizer(n_bits=8)
    q._delta, q._zero_float = torch.tensor(0.1), torch.tensor(3.0)
    y = torch.zeros(4, 8)
    idx = torch.zeros(4, 8, dtype=torch.int8)
    provenance.tag(y, q, idx)
    assert provenance.quantizer_of(y) is q and provenance.indices_of(y) is idx
    assert provenance.of(y.view(8, 4)) is None and provenance.of(y.clone()) is None and provenance.of(y + 0) is None
    y.add_(1.0)                                  # in-place: same object, other values
    assert provenance.of(y) is None
    provenance.tag(y, q, idx)
    assert provenance.of(y) is not None
    q._delta = torch.tensor(0.2)                 # producer's grid moved after the record was made
    assert provenance.of(y) is None
    provenance.tag(y, q, idx)
    q._delta.mul_(2.0)                           # ... or was updated in place (learned ranges)
    assert provenance.of(y) is None
    z = torch.ones(3)
    provenance.tag(z, q)
    key = id(z)
    del z
    assert key not in provenance._records        # records die with their tensor


def test_drop_in_equals_the_reference_on_random_networks(tmp_path):
    """Build container only: oracle/fuzz_models.py -- ONE script, run once against the reference's `quantization`
    package and once against this repository's (oracle-backed backend) -- on random small networks and quantization
    settings: every calibrated range, quantizer parameter and output (estimating, fixed-range, train-mode forwards) must
    be bit-identical, and where the reference raises, the same exception type must be raised here."""
    import os
    import subprocess
    import sys
    from tests.conftest import ROOT
    if not os.path.isdir('/root/reference/quantization'):
        pytest.skip('no /root/reference here (GPU box)')
    outs = {}
    # 'mine-inplace': the same with options.INPLACE_CALIBRATION_STATE (state buffers updated in place: the hipGraph-
    # capturable form of calibration) -- must not change a bit either
    # '*-double': the network family in float64 (`--double`, main.py:227-231)
    for impl in ('ref', 'mine', 'mine-inplace', 'ref-double', 'mine-double'):
        outs[impl] = str(tmp_path / f'{impl}.npz')
        extra = {'inplace': ['--inplace-state'], 'double': ['--double']}.get(impl.split('-')[-1], [])
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'fuzz_models.py'), '--impl', impl.split('-')[0],
                            '--n', '10', '--seed', '11', '--out', outs[impl]] + extra, capture_output=True, text=True,
                           cwd=str(tmp_path), timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    for ref_name, other in (('ref', 'mine'), ('ref', 'mine-inplace'), ('ref-double', 'mine-double')):
        a, b = np.load(outs[ref_name]), np.load(outs[other])
        assert set(a.keys()) == set(b.keys()), sorted(set(a.keys()) ^ set(b.keys()))[:10]
        n_ok = 0
        for k in a.keys():
            if k.endswith('_cfg'):
                continue
            if a[k].dtype.kind in 'US':
                assert str(a[k]) == str(b[k]), (other, k, str(a[k]), str(b[k]))
            else:
                assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=True), (other, k)
                n_ok += 1
        assert n_ok > (200 if 'double' not in other else 80)          # most cases are valid, with 10-40 observables each


def test_a_recorded_graph_keeps_the_derived_cache_tensors_it_reads_alive():
    """`derived_cache_tensors` finds every tensor the modules hold in a derived cache (the reference's cached quantized
    parameters, int8 weights, staircase tables, ...): GraphedForward / GraphedTrainStep store these references, so a
    cache rebuilt after capture cannot free memory a replay still reads by address."""
    import torch
    from torch import nn
    from quantization.graphs import derived_cache_tensors
    from quantization.autoquant_utils import QuantLinear
    net = nn.Sequential(QuantLinear(8, 8), nn.Sequential(QuantLinear(8, 4)))
    a, b, c, d = (torch.zeros(3) for _ in range(4))
    net[0].cached_params = (a, None)
    net[0]._int8_cache = (('key',), b, c)
    net[1][0]._int8_stair = {768: (('key', 768), (d, 768))}
    got = derived_cache_tensors(net)
    assert {id(t) for t in got} == {id(a), id(b), id(c), id(d)}
    assert derived_cache_tensors(nn.Linear(2, 2)) == []


def test_bench_refuses_a_stale_pmc_profile(tmp_path, monkeypatch):
    """`roofline.traffic` is a committed PMC profile, not a counter of the run: bench.py only quotes it when it was
    collected for this workload size, for the kernel symbol it measures and from the kernel sources the shipped library
    was built from."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('tq_bench_for_test', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = 1024 * 512 * 768
    good = {'workload_elems': n, 'kernel': 'void tq::fq_tensor<1, false, true, 4>(...)',
            'kernel_source_sha256': bench.kernel_source_hash(), 'traffic_bytes_per_launch': 1610704432}
    path = tmp_path / 'pmc_traffic.json'
    monkeypatch.setattr(bench, 'PMC_TRAFFIC_JSON', str(path))

    def write(**over):
        path.write_text(json.dumps({**good, **over}))
    write()
    assert bench.pmc_traffic(n)[0] == 1610704432
    write(kernel_source_sha256='0' * 64)
    v, why = bench.pmc_traffic(n)
    assert v is None and why.startswith('STALE')
    write(kernel='void tq::fq_axis<1>(...)')
    assert bench.pmc_traffic(n)[0] is None
    assert bench.pmc_traffic(n // 2)[0] is None
    path.unlink()
    assert bench.pmc_traffic(n)[0] is None


def test_inference_mode_keeps_the_layered_route_and_never_raises():
    """torch.inference_mode() tensors carry no version counter: the integer route ('auto' or forced) must stand down there
    instead of failing in its bookkeeping (provenance records, range-state keys, the golden-section memo)."""
    from quantization import _hip, options, provenance
    from quantization.base_quantized_classes import QuantizedActivation
    from quantization.quantizers import QMethods
    from quantization.range_estimators import OptMethod, RangeEstimators
    from tests._oracle_backend import OracleBackend
    prev = _hip.set_backend(OracleBackend())
    before = options.INT8_LINEAR
    try:
        for mode in ('auto', True):
            options.INT8_LINEAR = mode
            with torch.no_grad():
                assert options.int8_active()
            with torch.inference_mode():
                assert not options.int8_active()
                qa = QuantizedActivation(act_method=QMethods.asymmetric_uniform, n_bits_act=8,
                                         act_range_method=RangeEstimators.current_minmax)
                qa.quantized_acts()
                x = torch.randn(4, 8, 16)
                y0 = qa(x)                                   # estimating: ranges are set from inference tensors
                qa.activation_quantizer.fix_ranges()
                y = qa(x)
                assert torch.equal(y, y0)
                assert provenance.of(y) is None
                assert provenance.tag(y, qa.activation_quantizer.quantizer) is y and provenance.of(y) is None
                qa.activation_quantizer.quantizer.range_state_key()
                est = RangeEstimators.MSE.cls(quantizer=QMethods.symmetric_uniform.cls(n_bits=8), opt_method=OptMethod.golden_section)
                est(x)
                est._memoise(x)
                assert est._memo is None
    finally:
        options.INT8_LINEAR = before
        _hip.set_backend(prev)


def test_harness_constants_cached_under_inference_mode_do_not_reach_a_training_forward():
    """the per-shape helper tensors of the harness (position ids, token types, zero mask) are cached process-wide;
    one created under torch.inference_mode() is an inference tensor and `F.embedding(position ids)` of a later training
    forward with the same shape would refuse to save it for backward.  Inference-mode entries are kept apart."""
    from harness import bert
    bert._CONSTANTS.clear()
    with torch.inference_mode():
        p_inf = bert.constant('positions', 1, 7, 'cpu')
        assert p_inf.is_inference()
    p = bert.constant('positions', 1, 7, 'cpu')
    assert not p.is_inference() and torch.equal(p, p_inf)
    emb = torch.nn.Embedding(16, 4)
    emb(p).sum().backward()                                    # raised 'Inference tensors cannot be saved for backward'
    assert bert.constant('positions', 1, 7, 'cpu') is p         # still cached
    with torch.inference_mode():
        assert bert.constant('positions', 1, 7, 'cpu') is p_inf


@pytest.mark.default_route
def test_hooks_on_bypassed_containers_and_global_hooks_keep_the_layered_route():
    """a merged launch of the default route bypasses the __call__ of the CONTAINERS it replaces (BERT: the
    QResidualBlock `layer.output` and the Sequential around the intermediate Linear; MobileBERT: the two QBottleneckLayers,
    every QFFN / its QResidualNoNorm / its Sequential) -- a forward hook on one of them, or a global module hook, has to
    keep the modules that fire it.  Replayed on the CPU through the oracle backend."""
    from harness.bert import build_bert_base
    from harness.mobilebert import build_mobilebert
    from quantization import _hip, options
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from tests._oracle_backend import OracleBackend
    from utils.utils import pass_data_for_range_estimation
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
              weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)
    ids = torch.randint(0, 30000, (2, 64), generator=torch.Generator().manual_seed(0))
    prev = _hip.set_backend(OracleBackend())
    try:
        assert options.INT8_LINEAR == 'auto'
        for build, picks in ((build_bert_base, lambda m: [m.layers[0].output, m.layers[0].intermediate]),
                             (build_mobilebert, lambda m: [m.layers[0].bottleneck_attention, m.layers[0].ffn[1],
                                                           m.layers[0].ffn[0].output, m.layers[0].ffn[2].intermediate,
                                                           m.layers[0].output, m.layers[0].intermediate])):
            model, _ = build(seed=1000, num_layers=1, **qp)
            model = model.eval()
            with torch.no_grad():
                pass_data_for_range_estimation([(ids,)], model, act_quant=True, weight_quant=True, max_num_batches=1)
                model.fix_ranges()
                model(ids)                                            # the merged launches: no container is called
                for container in picks(model):
                    fired = []
                    h = container.register_forward_hook(lambda m, i, o: fired.append(1))
                    try:
                        model(ids)
                    finally:
                        h.remove()
                    assert fired == [1], (type(model).__name__, type(container).__name__, fired)
                seen = []
                h = torch.nn.modules.module.register_module_forward_hook(lambda m, i, o: seen.append(type(m).__name__))
                try:
                    model(ids)
                finally:
                    h.remove()
                # a global hook observes EVERY module call of the layered chain, e.g. each quantized Linear and each container
                n_lin = sum(1 for m in model.modules() if type(m).__name__ == 'QuantLinear')
                assert seen.count('QuantLinear') == n_lin, (seen.count('QuantLinear'), n_lin)
                assert any(n in seen for n in ('QResidualBlock', 'QResidualNoNorm'))
    finally:
        _hip.set_backend(prev)


@pytest.mark.default_route
def test_stacked_operand_caches_follow_in_place_parameter_updates():
    """the stacked NoNorm affine vectors of MobileBERT's grouped bottleneck launch and the stacked bias of the
    grouped Q | K | V launch are cached on the modules; an in-place update that keeps the pointers (optimizer step,
    load_state_dict's copy_) has to invalidate them.  CPU replay through the oracle backend: after the update the merged
    forward equals the un-merged integer forward again."""
    from harness.mobilebert import QMobileLayer, build_mobilebert
    from quantization import _hip, fused, options
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from tests._oracle_backend import OracleBackend
    from utils.utils import pass_data_for_range_estimation
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
              weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)
    ids = torch.randint(0, 30000, (2, 64), generator=torch.Generator().manual_seed(1))
    prev = _hip.set_backend(OracleBackend())
    try:
        model, _ = build_mobilebert(seed=1000, num_layers=1, **qp)
        model = model.eval()
        L = model.layers[0]
        with torch.no_grad():
            pass_data_for_range_estimation([(ids,)], model, act_quant=True, weight_quant=True, max_num_batches=1)
            model.fix_ranges()
            for nn_ in (L.bottleneck_input.LayerNorm, L.bottleneck_attention.LayerNorm):
                nn_._quant_w = False                                 # quantized_params() now returns the raw Parameters

            def unmerged():
                keep = fused.linear_nonorm_quant_pair, fused.quantized_self_attention, fused.quantized_ffn_chain
                fused.linear_nonorm_quant_pair = fused.quantized_self_attention = fused.quantized_ffn_chain = lambda *a, **k: None
                try:
                    return model(ids)
                finally:
                    fused.linear_nonorm_quant_pair, fused.quantized_self_attention, fused.quantized_ffn_chain = keep
            assert torch.equal(model(ids), unmerged())
            L.bottleneck_input.LayerNorm.weight.mul_(1.5)             # same storage, new values
            L.bottleneck_attention.LayerNorm.bias.add_(0.25)
            L.attention_self.key.bias.add_(0.5)                       # grouped query | key launch: stacked bias
            after = model(ids)
            assert torch.equal(after, unmerged())
    finally:
        _hip.set_backend(prev)
