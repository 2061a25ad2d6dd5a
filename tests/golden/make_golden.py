#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference, read-only).  The reference's
Python never travels to the GPU box; only the .npz data written here does.  Import recipe:
SURVEY.md section 8c (namespace shim for `utils`, no bytecode written).

    python tests/golden/make_golden.py

Every fixture is data only: inputs, configuration scalars and the reference's outputs.
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
REF = '/root/reference'
sys.path.insert(0, REF)
_u = types.ModuleType('utils')
_u.__path__ = [os.path.join(REF, 'utils')]
sys.modules['utils'] = _u

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch import nn  # noqa: E402

from quantization.quantizers import QMethods  # noqa: E402
from quantization.range_estimators import RangeEstimators, OptMethod  # noqa: E402
from quantization.quantization_manager import QuantizationManager  # noqa: E402
from quantization.base_quantized_model import QuantizedModel  # noqa: E402
from quantization.base_quantized_classes import QuantizedActivation  # noqa: E402
from quantization.autoquant_utils import quantize_model, QuantLinear  # noqa: E402
from quantization.adaround import apply_adaround_to_layer, AdaRoundMode  # noqa: E402
from quantization.adaround.config import DEFAULT_ADAROUND_CONFIG  # noqa: E402
from quantization.adaround.quantizer import ADAROUND_QUANTIZER_MAP  # noqa: E402
from quantization.adaround.utils import CombinedLoss, MODE_TO_LOSS_TYPE  # noqa: E402
from utils.per_embd_quant_utils import set_act_quant_axis_and_groups  # noqa: E402
from utils.utils import pass_data_for_range_estimation  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)


def hidden_like(shape, seed, outlier_dims=(5, 17), scale=20.0):
    """BERT-like hidden state: unit normal with a couple of outlier embedding dims
     and a x3 stronger outlier on the last token."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g)
    for d in outlier_dims:
        if d < shape[-1]:
            x[..., d] *= scale
            x[..., -1, d] *= 3.0
    return x


def np32(t):
    return t.detach().to(torch.float32).cpu().numpy().copy()  # copy: estimators update in place


def bf16_bits(t):
    return t.detach().to(torch.bfloat16).view(torch.int16).cpu().numpy().view(np.uint16)


# ----------------------------------------------------------------------------------- 1
def gen_fake_quant():
    """Fixed-range fake-quant: (x, range) -> delta, zero_float, signed, indices, y."""
    data, meta = {}, []
    k = 0
    B, T, D = 2, 16, 48
    for method in ('asymmetric_uniform', 'symmetric_uniform'):
        for n_bits in (4, 8, 16):
            for layout in ('per_tensor', 'per_embd', 'peg6', 'peg6_perm', 'per_channel'):
                if method == 'symmetric_uniform' and layout in ('per_embd', 'peg6', 'peg6_perm'):
                    continue  # reference crashes (quirk q3, quantizers.py:217)
                for io in ('fp32', 'bf16'):
                    for signed_data in (True, False):
                        if not signed_data and layout not in ('per_tensor', 'per_channel'):
                            continue
                        seed = 1000 + k
                        if layout == 'per_channel':
                            g = torch.Generator().manual_seed(seed)
                            x = torch.randn(24, 40, generator=g) * \
                                torch.linspace(0.2, 3.0, 24).view(-1, 1)
                        else:
                            x = hidden_like((B, T, D), seed)
                        if not signed_data:
                            x = x.abs()
                        if io == 'bf16':
                            x_store = x.to(torch.bfloat16)
                            x = x_store.float()
                        mgr = QuantizationManager(
                            qmethod=QMethods[method], init=RangeEstimators.current_minmax,
                            per_channel=(layout == 'per_channel'),
                            qparams=dict(n_bits=n_bits))
                        if layout == 'per_embd':
                            set_act_quant_axis_and_groups(mgr, axis=2, n_groups=None)
                        elif layout == 'peg6':
                            set_act_quant_axis_and_groups(mgr, axis=2, n_groups=6)
                        elif layout == 'peg6_perm':
                            set_act_quant_axis_and_groups(mgr, axis=2, n_groups=6, permute=True)
                            mgr(x)  # phase 1: collect ranges
                            data[f'c{k}_ranges'] = np32(mgr.range_estimator.ranges)
                            mgr.range_estimator.per_group_range_estimation = False
                        y = mgr(x)
                        mgr.fix_ranges()
                        y2 = mgr(x)
                        assert torch.equal(y, y2)
                        q = mgr.quantizer
                        idx = q.to_integer_forward(x)
                        data[f'c{k}_x'] = bf16_bits(x) if io == 'bf16' else np32(x)
                        data[f'c{k}_xmin'] = np32(mgr.range_estimator.current_xmin)
                        data[f'c{k}_xmax'] = np32(mgr.range_estimator.current_xmax)
                        data[f'c{k}_delta'] = np32(q._delta)
                        if q._zero_float is not None:
                            data[f'c{k}_zero_float'] = np32(q._zero_float)
                        data[f'c{k}_idx'] = np32(idx)
                        data[f'c{k}_y'] = np32(y)
                        if io == 'bf16':
                            data[f'c{k}_y_bf16'] = bf16_bits(y)
                        meta.append(dict(
                            k=k, method=method, n_bits=n_bits, layout=layout, io=io,
                            signed=(bool(q.signed) if method == 'symmetric_uniform' else None),
                            int_min=float(q.int_min), int_max=float(q.int_max)))
                        k += 1
    data['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, 'fake_quant.npz'), **data)
    print('fake_quant cases:', k)


# ----------------------------------------------------------------------------------- 2
def gen_estimators():
    """3-batch traces of every range estimator (+ the quantised output of batch 3)."""
    data, meta = {}, []
    k = 0
    B, T, D = 2, 12, 24
    batches = [hidden_like((B, T, D), 2000 + i, outlier_dims=(3, 11)) * (1.0 + 0.3 * i)
               for i in range(3)]
    data['batches'] = np.stack([np32(b) for b in batches])
    wbatches = [torch.randn(6, 20, generator=torch.Generator().manual_seed(2100 + i))
                for i in range(3)]
    data['wbatches'] = np.stack([np32(b) for b in wbatches])
    pos_batches = [b.abs() for b in batches]
    # larger inputs (round 2): a weight-shaped [64, 96] matrix (rows > 8: the second torch.sum of loss_fx runs
    # ATen's vector path) and a [4, 40, 96] activation (3840-element rows: cascade levels of the row sums)
    wbig = [torch.randn(64, 96, generator=torch.Generator().manual_seed(2300 + i)) * 0.05 * (1 + i)
            for i in range(3)]
    data['wbig'] = np.stack([np32(b) for b in wbig])
    abig = [hidden_like((4, 40, 96), 2400 + i, outlier_dims=(7, 40)) for i in range(3)]
    data['abig'] = np.stack([np32(b) for b in abig])

    def run(name, method, init, n_bits, layout, init_params, inputs='batches'):
        nonlocal k
        mgr = QuantizationManager(
            qmethod=QMethods[method], init=RangeEstimators[init],
            per_channel=(layout == 'per_channel'), qparams=dict(n_bits=n_bits),
            init_params=dict(init_params))
        if layout == 'per_embd':
            set_act_quant_axis_and_groups(mgr, axis=2, n_groups=None)
        elif layout == 'peg4':
            set_act_quant_axis_and_groups(mgr, axis=2, n_groups=4)
        xs = {'batches': batches, 'wbatches': wbatches, 'pos_batches': pos_batches, 'wbig': wbig,
              'abig': abig}[inputs]
        mins, maxs, deltas, zfs = [], [], [], []
        for x in xs:
            y = mgr(x)
            mins.append(np32(mgr.range_estimator.current_xmin).reshape(-1))
            maxs.append(np32(mgr.range_estimator.current_xmax).reshape(-1))
            deltas.append(np32(mgr.quantizer._delta).reshape(-1))
            if mgr.quantizer._zero_float is not None:
                zfs.append(np32(mgr.quantizer._zero_float).reshape(-1))
        data[f'e{k}_xmin'] = np.stack(mins)
        data[f'e{k}_xmax'] = np.stack(maxs)
        data[f'e{k}_delta'] = np.stack(deltas)
        if zfs:
            data[f'e{k}_zero_float'] = np.stack(zfs)
        data[f'e{k}_y_last'] = np32(y)
        est = mgr.range_estimator
        if getattr(est, 'loss_array', None) is not None:
            data[f'e{k}_loss_array'] = np.asarray(est.loss_array, dtype=np.float64)
        ip = {kk: (vv.name if isinstance(vv, OptMethod) else vv) for kk, vv in init_params.items()}
        meta.append(dict(k=k, name=name, method=method, init=init, n_bits=n_bits, layout=layout,
                         init_params=ip, inputs=inputs,
                         signed=(bool(mgr.quantizer.signed)
                                 if method == 'symmetric_uniform' else None)))
        k += 1

    for method in ('asymmetric_uniform', 'symmetric_uniform'):
        for init in ('current_minmax', 'allminmax', 'running_minmax'):
            run(f'{init}-tensor', method, init, 8, 'per_tensor', {})
            run(f'{init}-channel', method, init, 8, 'per_channel', {}, inputs='wbatches')
    for init in ('current_minmax', 'running_minmax', 'allminmax'):
        run(f'{init}-embd', 'asymmetric_uniform', init, 8, 'per_embd', {})
        run(f'{init}-peg4', 'asymmetric_uniform', init, 8, 'peg4', {})
    run('running-m0.5', 'asymmetric_uniform', 'running_minmax', 4, 'per_tensor',
        dict(momentum=0.5))
    # MSE: 1-D grid (symmetric; one-sided asymmetric), 2-D grid (two-sided asymmetric, small)
    run('mse1d-sym', 'symmetric_uniform', 'MSE', 8, 'per_tensor', dict(num_candidates=100))
    run('mse1d-sym4', 'symmetric_uniform', 'MSE', 4, 'per_tensor', dict(num_candidates=50))
    run('mse1d-onesided', 'asymmetric_uniform', 'MSE', 8, 'per_tensor',
        dict(num_candidates=100), inputs='pos_batches')
    run('mse1d-sym-channel', 'symmetric_uniform', 'MSE', 4, 'per_channel',
        dict(num_candidates=40), inputs='wbatches')
    run('mse2d-asym4', 'asymmetric_uniform', 'MSE', 4, 'per_tensor', dict(num_candidates=20))
    run('mse2d-asym6', 'asymmetric_uniform', 'MSE', 6, 'per_tensor', dict(num_candidates=12))
    run('mse-peg-degenerate', 'asymmetric_uniform', 'MSE', 4, 'peg4', dict(num_candidates=10))
    run('golden-sym', 'symmetric_uniform', 'MSE', 8, 'per_tensor',
        dict(opt_method=OptMethod.golden_section))
    run('golden-sym-channel', 'symmetric_uniform', 'MSE', 4, 'per_channel',
        dict(opt_method=OptMethod.golden_section), inputs='wbatches')
    run('golden-asym', 'asymmetric_uniform', 'MSE', 4, 'per_tensor',
        dict(opt_method=OptMethod.golden_section))

    # cross-entropy estimator on logits [B, num_labels]
    g = torch.Generator().manual_seed(2200)
    logits = [torch.randn(8, 3, generator=g) * 2.0 for _ in range(3)]
    data['logits'] = np.stack([np32(b) for b in logits])
    for method in ('asymmetric_uniform', 'symmetric_uniform'):
        mgr = QuantizationManager(qmethod=QMethods[method], init=RangeEstimators.cross_entropy,
                                  qparams=dict(n_bits=4), init_params=dict(num_candidates=16))
        mins, maxs = [], []
        for x in logits:
            y = mgr(x)
            mins.append(np32(mgr.range_estimator.current_xmin).reshape(-1))
            maxs.append(np32(mgr.range_estimator.current_xmax).reshape(-1))
        data[f'e{k}_xmin'] = np.stack(mins)
        data[f'e{k}_xmax'] = np.stack(maxs)
        data[f'e{k}_y_last'] = np32(y)
        data[f'e{k}_loss_array'] = np.asarray(mgr.range_estimator.loss_array, dtype=np.float64)
        meta.append(dict(k=k, name='xent', method=method, init='cross_entropy', n_bits=4,
                         layout='per_tensor', init_params=dict(num_candidates=16),
                         inputs='logits',
                         signed=(bool(mgr.quantizer.signed)
                                 if method == 'symmetric_uniform' else None)))
        k += 1

    # round 2 (appended so that earlier trace numbers stay put): the full asymmetric 8-bit 2-D grid
    # (100 x 64 x 2 candidates, range_estimators.py:378-420) and golden-section searches on larger inputs,
    # incl. the README's weight recipe shape class (MSE / golden_section, symmetric, per-tensor; README.md:149-157)
    run('mse2d-asym8-full', 'asymmetric_uniform', 'MSE', 8, 'per_tensor', dict(num_candidates=100))
    run('golden-asym8', 'asymmetric_uniform', 'MSE', 8, 'per_tensor',
        dict(opt_method=OptMethod.golden_section))
    run('golden-sym8-weight', 'symmetric_uniform', 'MSE', 8, 'per_tensor',
        dict(opt_method=OptMethod.golden_section), inputs='wbig')
    run('golden-sym4-weight-channel', 'symmetric_uniform', 'MSE', 4, 'per_channel',
        dict(opt_method=OptMethod.golden_section), inputs='wbig')
    run('golden-asym8-act', 'asymmetric_uniform', 'MSE', 8, 'per_tensor',
        dict(opt_method=OptMethod.golden_section), inputs='abig')
    run('mse1d-sym8-weight-channel', 'symmetric_uniform', 'MSE', 8, 'per_channel',
        dict(num_candidates=100), inputs='wbig')
    run('mse2d-asym8-act', 'asymmetric_uniform', 'MSE', 8, 'per_tensor', dict(num_candidates=30),
        inputs='abig')

    # permuted PEG: phase 1 (ranges) then group statistics
    mgr = QuantizationManager(qmethod=QMethods.asymmetric_uniform,
                              init=RangeEstimators.current_minmax, qparams=dict(n_bits=8))
    set_act_quant_axis_and_groups(mgr, axis=2, n_groups=4, permute=True)
    for x in batches:
        out = mgr(x)
        assert out is x
    data['perm_ranges'] = np32(mgr.range_estimator.ranges)
    mgr.range_estimator.per_group_range_estimation = False
    y = mgr(batches[0])
    data['perm_xmin'] = np32(mgr.range_estimator.current_xmin)
    data['perm_xmax'] = np32(mgr.range_estimator.current_xmax)
    data['perm_y'] = np32(y)

    # Oracle of the per-group MSE EXTENSION: the reference's own per-channel MSE
    # estimator applied to the [n_groups, -1] view of each batch, then repeat_interleave(gs).
    from quantization.quantizers import AsymmetricUniformQuantizer, SymmetricUniformQuantizer
    ng = 4
    for tag, qcls, n_bits, inputs_, ip_ in (('asym4', AsymmetricUniformQuantizer, 4, batches, dict(num_candidates=12)),
                                            ('sym8', SymmetricUniformQuantizer, 8, batches, dict(num_candidates=50)),
                                            ('onesided6', AsymmetricUniformQuantizer, 6, pos_batches, dict(num_candidates=30))):
        qz = qcls(n_bits=n_bits)
        est = RangeEstimators.MSE.cls(per_channel=True, quantizer=qz, **ip_)
        mins, maxs = [], []
        for x in inputs_:
            xg = x.transpose(0, 2).contiguous().view(x.size(2), -1).view(ng, -1)
            mn, mx = est(xg)
            gs = x.size(2) // ng
            mins.append(np32(mn.repeat_interleave(gs)))
            maxs.append(np32(mx.repeat_interleave(gs)))
        data[f'pg_{tag}_xmin'] = np.stack(mins)
        data[f'pg_{tag}_xmax'] = np.stack(maxs)
        data[f'pg_{tag}_loss'] = np.asarray(est.loss_array, dtype=np.float64)

    data['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, 'estimators.npz'), **data)
    print('estimator cases:', k)


# ----------------------------------------------------------------------------------- 3
class ToyNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(24, 32)
        self.act = nn.GELU()
        self.ln = nn.LayerNorm(32)
        self.fc2 = nn.Linear(32, 24)

    def forward(self, x):
        return self.fc2(self.ln(self.act(self.fc1(x))))


class QuantToy(QuantizedModel):
    def __init__(self, org, **qp):
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


def gen_toy_model():
    """Calibration trace of a 2-layer QuantizedModel through pass_data_for_range_estimation."""
    torch.manual_seed(3000)
    org = ToyNet()
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform,
              n_bits=8, n_bits_act=8, weight_range_method=RangeEstimators.current_minmax,
              act_range_method=RangeEstimators.running_minmax)
    model = QuantToy(org, **qp)
    loader = [(hidden_like((2, 12, 24), 3100 + i, outlier_dims=(3, 11)),) for i in range(4)]
    pass_data_for_range_estimation(loader, model, act_quant=True, weight_quant=True,
                                   max_num_batches=3)
    model.fix_ranges()
    model.eval()
    out = model(loader[3][0])
    data = {'loader': np.stack([np32(b[0]) for b in loader]), 'out': np32(out)}
    for kname, v in org.state_dict().items():
        data['w_' + kname] = np32(v)
    sd = model.state_dict()
    names = []
    for kname, v in sd.items():
        if any(s in kname for s in ('_delta', '_zero_float', '_signed', 'current_x')):
            data['sd_' + kname] = v.detach().cpu().numpy()
            names.append(kname)
    data['sd_names'] = np.array(json.dumps(names))
    np.savez_compressed(os.path.join(OUT, 'toy_model.npz'), **data)
    print('toy model state entries:', len(names))


# ----------------------------------------------------------------------------------- 4
def gen_adaround():
    """AdaRound: alpha init, soft/hard W_q, and an N-step optimisation trace on a tiny
    QuantLinear with a recorded batch-index sequence (neutralises quirk q11)."""
    data, meta = {}, []
    k = 0
    for method in ('symmetric_uniform', 'asymmetric_uniform'):
        for mode in ('learned_hard_sigmoid', 'learned_sigmoid', 'sigmoid_temp_decay'):
            torch.manual_seed(4000 + k)
            lin = nn.Linear(16, 12)
            layer = QuantLinear(16, 12, method=QMethods[method], n_bits=4,
                                weight_range_method=RangeEstimators.current_minmax)
            layer.weight.data = lin.weight.data.clone()
            layer.bias.data = lin.bias.data.clone()
            layer.quantized_weights()
            layer.caching = False
            X = torch.randn(32, 6, 16)
            with torch.no_grad():
                layer.full_precision()
                tgt = layer(X)
                layer.quantized_weights()
                layer(X[:4])  # initialises the weight range (current min/max)
            oq = layer.weight_quantizer.quantizer
            cls = ADAROUND_QUANTIZER_MAP[oq.__class__]
            wq = cls(n_bits=oq.n_bits, scale_domain=oq.scale_domain, per_channel=oq.per_channel,
                     eps=oq.eps)
            wq.register_buffer('_delta', oq._delta)
            wq.register_buffer('_zero_float', oq._zero_float)
            if hasattr(oq, '_signed'):
                wq.register_buffer('_signed', oq._signed)
            layer.weight_quantizer.quantizer = wq
            layer.weight_quantizer.fix_ranges()
            wq.round_mode = AdaRoundMode[mode]
            wq.temperature = 20
            wq.soft_targets = True
            with torch.no_grad():
                wq_soft0 = wq(layer.weight)         # inits alpha
            alpha0 = wq.alpha.detach().clone()
            wq.soft_targets = False
            with torch.no_grad():
                wq_hard0 = wq(layer.weight)
                idx_hard0 = wq.to_integer_forward(layer.weight)
            wq.soft_targets = True
            iters, bs = 12, 4
            loss_fn = CombinedLoss(quantizer=wq, loss_type=MODE_TO_LOSS_TYPE[wq.round_mode],
                                   weight=0.01, max_count=iters, b_range=(20, 2), warmup=0.2,
                                   decay_type=DEFAULT_ADAROUND_CONFIG.decay_type,
                                   decay_shape=1.0, decay_start=0.0)
            opt = torch.optim.Adam([wq.alpha], lr=1e-2)
            g = torch.Generator().manual_seed(4100 + k)
            idxs, losses, alphas, grads = [], [], [], []
            for it in range(iters):
                idx = torch.randperm(X.size(0), generator=g)[:bs]
                idxs.append(idx.numpy())
                opt.zero_grad()
                out = layer(X[idx])
                loss = loss_fn(out, tgt[idx])
                loss.backward()
                grads.append(np32(wq.alpha.grad))
                opt.step()
                losses.append(float(loss))
                alphas.append(np32(wq.alpha))
            wq.soft_targets = False
            with torch.no_grad():
                wq_hard1 = wq(layer.weight)
            data[f'a{k}_w'] = np32(layer.weight)
            data[f'a{k}_b'] = np32(layer.bias)
            data[f'a{k}_X'] = np32(X)
            data[f'a{k}_tgt'] = np32(tgt)
            data[f'a{k}_delta'] = np32(wq._delta)
            if wq._zero_float is not None:
                data[f'a{k}_zero_float'] = np32(wq._zero_float)
            data[f'a{k}_alpha0'] = np32(alpha0)
            data[f'a{k}_wq_soft0'] = np32(wq_soft0)
            data[f'a{k}_wq_hard0'] = np32(wq_hard0)
            data[f'a{k}_idx_hard0'] = np32(idx_hard0)
            data[f'a{k}_batch_idx'] = np.stack(idxs)
            data[f'a{k}_losses'] = np.array(losses, dtype=np.float64)
            data[f'a{k}_alphas'] = np.stack(alphas)
            data[f'a{k}_grads'] = np.stack(grads)
            data[f'a{k}_wq_hard1'] = np32(wq_hard1)
            meta.append(dict(k=k, method=method, mode=mode, n_bits=4, iters=iters, bs=bs, lr=1e-2,
                             signed=(bool(wq.signed) if method == 'symmetric_uniform' else None),
                             int_min=float(wq.int_min), int_max=float(wq.int_max)))
            k += 1
    data['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, 'adaround.npz'), **data)
    print('adaround cases:', k)


# ----------------------------------------------------------------------------------- 5
def gen_adaround_inits():
    """AdaRound grid inits (adaround/adaround.py:160-201) and whole apply_adaround_to_layer runs on the 2-layer toy
    QuantizedModel, plus sampled values of the annealing schedule (adaround/utils.py:93-128)."""
    import copy
    from quantization.adaround.adaround import apply_mse_init, apply_mse_out_init
    from quantization.adaround.utils import (TempDecay, AdaRoundTempDecayType, AdaRoundInitMode, GetLayerInpOut,
                                            LayerOutputMSE)
    data, meta = {}, []
    torch.manual_seed(5000)
    org = ToyNet()
    for kname, v in org.state_dict().items():
        data['w_' + kname] = np32(v)
    X = hidden_like((32, 12, 24), 5100, outlier_dims=(3, 11))
    data['X'] = np32(X)
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=4, n_bits_act=8,
              weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)

    def fresh():
        m = QuantToy(copy.deepcopy(org), **qp)
        m.eval()
        m.set_quant_state(weight_quant=True, act_quant=False)
        with torch.no_grad():
            m(X[:8])                                   # initialises the weight ranges
        m.fix_ranges() if False else None
        for mod in m.modules():
            if isinstance(mod, QuantizationManager) and mod.quantizer.is_initialized:
                mod.fix_ranges()
        return m

    k = 0
    for lname in ('fc1', 'fc2'):
        # (a) weight-MSE init
        m = fresh()
        layer = getattr(m, lname)
        d0 = np32(layer.weight_quantizer.quantizer._delta)
        apply_mse_init(layer)
        data[f'i{k}_delta0'] = d0
        data[f'i{k}_delta_mse'] = np32(layer.weight_quantizer.quantizer._delta)
        # (b) output-MSE inits, with the 80 scores the reference's LayerOutputMSE yields
        for asym in (False, True):
            m = fresh()
            layer = getattr(m, lname)
            layer.caching = False
            w = layer.weight
            q = layer.weight_quantizer.quantizer
            loss_fn = LayerOutputMSE(layer, GetLayerInpOut(m, layer, asym=asym), X, 8)
            scores = []
            with torch.no_grad():
                w_absmax = torch.max(w.max(), torch.abs(w.min()))
                for i in range(80):
                    sv = w_absmax * (1.0 - 0.01 * i)
                    q.set_quant_range(-sv, sv)
                    scores.append(loss_fn())
            m = fresh()
            layer = getattr(m, lname)
            layer.caching = False
            apply_mse_out_init(m, layer, X, 8, asym=asym)
            tag = 'asym' if asym else 'sym'
            data[f'i{k}_scores_out_{tag}'] = np.array(scores, dtype=np.float64)
            data[f'i{k}_delta_out_{tag}'] = np32(layer.weight_quantizer.quantizer._delta)
        # (c) whole-layer AdaRound runs (global torch RNG seeded right before, quirk q11)
        for init in ('range_estimator', 'mse', 'mse_out'):
            m = fresh()
            layer = getattr(m, lname)
            cfg = copy.deepcopy(DEFAULT_ADAROUND_CONFIG)
            cfg.iters = 40
            cfg.lr = 1e-2
            cfg.init = AdaRoundInitMode[init]
            m.full_precision()
            layer.quantized_weights()
            torch.manual_seed(5200 + k)
            res = apply_adaround_to_layer(m, layer, X, batch_size=8, act_quant=False, adaround_config=cfg)
            wq = layer.weight_quantizer.quantizer
            with torch.no_grad():
                hard = wq(layer.weight)
            data[f'i{k}_{init}_alpha'] = np32(wq.alpha)
            data[f'i{k}_{init}_hard'] = np32(hard)
            data[f'i{k}_{init}_delta'] = np32(wq._delta)
            data[f'i{k}_{init}_losses'] = np.array([res.loss_soft_before, res.loss_hard_before, res.loss_soft_after,
                                                    res.loss_hard_after], dtype=np.float64)
        meta.append(dict(k=k, layer=lname, iters=40, lr=1e-2, bs=8, seed=5200 + k))
        k += 1
    # annealing schedule samples
    sched = {}
    for name in ('linear', 'cosine', 'sigmoid', 'power', 'exp', 'log'):
        for shape in (1.0, 2.5):
            td = TempDecay(1000, (20, 2), 0.2, AdaRoundTempDecayType[name], shape)
            sched[f'{name}_{shape}'] = [float(td(t)) for t in range(0, 1001, 25)]
    data['schedule'] = np.array(json.dumps(sched))
    data['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, 'adaround_inits.npz'), **data)
    print('adaround init cases:', k)


# ----------------------------------------------------------------------------------- 6
def gen_double():
    """`--double` (main.py:227-231): the reference's QuantizationManager (quantizer + estimator) on float64 tensors,
    2-3 calibration batches then a fixed-range forward.  Everything stored as float64."""
    cases = [
        # (method, estimator, per_channel, axis, n_bits, shape, batches)
        ('asymmetric_uniform', 'current_minmax', False, -1, 8, (4, 24, 48), 2),
        ('asymmetric_uniform', 'running_minmax', False, -1, 8, (4, 24, 48), 3),
        ('asymmetric_uniform', 'allminmax', False, -1, 6, (4, 24, 48), 3),
        ('symmetric_uniform', 'current_minmax', False, -1, 8, (64, 96), 1),
        ('symmetric_uniform', 'current_minmax', True, -1, 4, (64, 96), 1),
        ('asymmetric_uniform', 'current_minmax', True, -1, 8, (32, 50), 1),
        ('asymmetric_uniform', 'running_minmax', False, 2, 8, (4, 24, 48), 3),
        ('asymmetric_uniform', 'current_minmax', False, 2, 4, (2, 16, 64), 2),
        ('symmetric_uniform', 'MSE', False, -1, 4, (48, 64), 1),
        ('asymmetric_uniform', 'MSE', False, -1, 4, (48, 64), 2),
    ]
    data = {'n_cases': np.array(len(cases))}
    for i, (method, est, per_channel, axis, n_bits, shape, nb) in enumerate(cases):
        qparams = dict(n_bits=n_bits)
        kw = dict(axis=axis) if axis >= 0 else {}
        init_params = dict(num_candidates=20) if est == 'MSE' else {}
        m = QuantizationManager(QMethods[method], init=RangeEstimators[est], per_channel=per_channel, qparams=qparams,
                                init_params=init_params, **kw)
        m.estimate_ranges()
        xs = torch.stack([hidden_like(shape, 7000 + 10 * i + b).double() * (1.0 + 0.3 * b) for b in range(nb)])
        with torch.no_grad():
            for b in range(nb):
                y = m(xs[b])
            assert y.dtype == torch.float64
            m.fix_ranges()
            y_fixed = m(xs[0])
        data[f'c{i}_method'] = np.array(method)
        data[f'c{i}_estimator'] = np.array(est)
        data[f'c{i}_per_channel'] = np.array(per_channel)
        data[f'c{i}_axis'] = np.array(axis)
        data[f'c{i}_n_bits'] = np.array(n_bits)
        data[f'c{i}_x'] = xs.numpy()
        data[f'c{i}_delta'] = m.quantizer._delta.detach().numpy().copy()       # float64, or float32 after MSE (python floats)
        if getattr(m.quantizer, '_zero_float', None) is not None:
            data[f'c{i}_zero_float'] = m.quantizer._zero_float.detach().numpy().copy()
        data[f'c{i}_y'] = y.numpy()
        data[f'c{i}_y_fixed'] = y_fixed.numpy()
        print(i, method, est, 'delta dtype', data[f'c{i}_delta'].dtype)
    np.savez_compressed(os.path.join(OUT, 'double.npz'), **data)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'double':
        gen_double()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'adaround_inits':
        gen_adaround_inits()
        sys.exit(0)
    gen_fake_quant()
    gen_estimators()
    gen_toy_model()
    gen_adaround()
    gen_adaround_inits()
    for f in sorted(os.listdir(OUT)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, 'KiB')
