#!/usr/bin/env python3
"""Golden vectors for the MODEL-level AdaRound driver: the reference's `apply_adaround_to_model`
(utils/adaround_utils.py:35-139) with `layers=('all',)` on a small QuantizedModel (Linear -> ReLU -> LayerNorm ->
Linear, W4A8), driven the way main.py:560-582 drives it -- calibrate + fix ranges, then AdaRound layer after layer
(asymmetric reconstruction: layer k sees the rounded layers < k), then (`post_adaround`) a fresh activation calibration
on top of the learned weights -- by IMPORTING the reference (build container only; the .npz is data).

    python tests/golden/make_golden_adaround_model.py

Weights and samples come from numpy's legacy Mersenne-Twister stream (bit-stable across numpy versions / platforms), so
the test regenerates them instead of trusting a torch RNG of another build; the sample indices of every iteration come
from torch's global RNG (`torch.randperm`, quirk q11), seeded right before the call, and are additionally recorded.
"""
import copy
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
from quantization.range_estimators import RangeEstimators  # noqa: E402
from quantization.quantization_manager import QuantizationManager  # noqa: E402
from quantization.base_quantized_model import QuantizedModel  # noqa: E402
from quantization.autoquant_utils import quantize_model  # noqa: E402
from quantization.adaround.config import DEFAULT_ADAROUND_CONFIG  # noqa: E402
from utils.adaround_utils import apply_adaround_to_model  # noqa: E402
from utils.utils import DotDict, pass_data_for_range_estimation  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
D_IN, D_MID, D_OUT, T = 24, 48, 16, 6
W_SEED, X_SEED, RNG_SEED = 9400, 9401, 9402
N_SAMPLES, BS, ITERS, LR = 32, 8, 30, 1e-2


def weights():
    rs = np.random.RandomState(W_SEED)
    w1 = (rs.standard_normal((D_MID, D_IN)) * 0.3).astype(np.float32)
    b1 = (rs.standard_normal(D_MID) * 0.1).astype(np.float32)
    ln_w = (1.0 + 0.25 * rs.standard_normal(D_MID)).astype(np.float32)
    ln_b = (0.05 * rs.standard_normal(D_MID)).astype(np.float32)
    w2 = (rs.standard_normal((D_OUT, D_MID)) * 0.2).astype(np.float32)
    b2 = (rs.standard_normal(D_OUT) * 0.1).astype(np.float32)
    return w1, b1, ln_w, ln_b, w2, b2


def samples():
    rs = np.random.RandomState(X_SEED)
    x = rs.standard_normal((N_SAMPLES, T, D_IN)).astype(np.float32)
    x[..., 3] *= 6.0                                 # one outlier dimension, as in BERT's hidden states
    return x


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(D_IN, D_MID)
        self.act = nn.ReLU()
        self.ln = nn.LayerNorm(D_MID, eps=1e-12)
        self.fc2 = nn.Linear(D_MID, D_OUT)
        with torch.no_grad():
            for p, w in zip((self.fc1.weight, self.fc1.bias, self.ln.weight, self.ln.bias, self.fc2.weight, self.fc2.bias),
                            weights()):
                p.copy_(torch.from_numpy(w))


class QNet(QuantizedModel):
    def __init__(self, org, **qp):
        super().__init__()
        self.fc1 = quantize_model(nn.Sequential(org.fc1, org.act), **qp)[0]      # Linear with the ReLU folded in
        self.ln = quantize_model(org.ln, **qp)
        self.fc2 = quantize_model(org.fc2, **qp)

    def forward(self, x):
        return self.fc2(self.ln(self.fc1(x)))


def main():
    torch.set_num_threads(8)
    x = torch.from_numpy(samples())
    loader = [(x[i:i + BS],) for i in range(0, N_SAMPLES, BS)]
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=4, n_bits_act=8,
              weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)
    model = QNet(Net(), **qp)
    model.eval()
    config = DotDict(quant=DotDict(act_quant=True, weight_quant=True),
                     act_quant=DotDict(num_batches=2, cross_entropy_layer=None),
                     adaround=copy.deepcopy(DEFAULT_ADAROUND_CONFIG))
    config.adaround.iters, config.adaround.lr, config.adaround.num_samples = ITERS, LR, N_SAMPLES
    # main.py:243-266: estimate + fix ranges, then set the quant state
    pass_data_for_range_estimation(loader=loader, model=model, act_quant=True, weight_quant=True,
                                   max_num_batches=config.act_quant.num_batches,
                                   cross_entropy_layer=config.act_quant.cross_entropy_layer)
    model.fix_ranges()
    model.set_quant_state(weight_quant=True, act_quant=True)
    data = {}
    with torch.no_grad():
        data['out_before'] = model(x[:BS]).numpy()

    # record the sample indices the reference draws (torch.randperm on the global RNG), without disturbing them
    drawn = []
    real_randperm = torch.randperm

    def spy(n, *a, **k):
        r = real_randperm(n, *a, **k)
        drawn.append(r[:BS].clone().numpy())
        return r
    torch.randperm = spy
    torch.manual_seed(RNG_SEED)
    try:
        apply_adaround_to_model(config, model, data_loader=loader, range_est_data_loader=loader, batch_size=BS)
    finally:
        torch.randperm = real_randperm
    data['drawn'] = np.stack(drawn).astype(np.int64)

    names = []
    for name, mod in model.named_modules():
        if hasattr(mod, 'weight_quantizer') and hasattr(mod, 'weight'):
            wq = mod.weight_quantizer.quantizer
            names.append(name)
            with torch.no_grad():
                idx = wq.to_integer_forward(mod.weight)
                wq_hard = wq(mod.weight)
            data[f'{name}_delta'] = wq._delta.detach().numpy().astype(np.float32).reshape(-1)
            data[f'{name}_alpha'] = wq.alpha.detach().numpy().astype(np.float32)
            data[f'{name}_hard_idx'] = idx.numpy().astype(np.float32)
            data[f'{name}_w_q'] = wq_hard.numpy().astype(np.float32)
            data[f'{name}_soft_targets'] = np.array(bool(wq.soft_targets))
    act = [(n, m) for n, m in model.named_modules()
           if isinstance(m, QuantizationManager) and n.endswith('activation_quantizer')]
    data['act_names'] = np.array([n for n, _ in act])
    data['act_min'] = np.array([float(m.range_estimator.current_xmin) for _, m in act], np.float32)
    data['act_max'] = np.array([float(m.range_estimator.current_xmax) for _, m in act], np.float32)
    data['act_delta'] = np.array([float(m.quantizer._delta) for _, m in act], np.float32)
    data['act_state'] = np.array([m.state.name for _, m in act])
    data['w_state'] = np.array([getattr(model, n).weight_quantizer.state.name for n in names])
    with torch.no_grad():
        data['out_after'] = model(x[:BS]).numpy()
    data['sd_keys'] = np.array(list(model.state_dict().keys()))
    data['meta'] = np.array(json.dumps(dict(
        layers=names, d_in=D_IN, d_mid=D_MID, d_out=D_OUT, t=T, w_seed=W_SEED, x_seed=X_SEED, rng_seed=RNG_SEED,
        n_samples=N_SAMPLES, bs=BS, iters=ITERS, lr=LR, n_bits=4, quant_act_after=bool(config.quant.act_quant),
        versions=dict(torch=torch.__version__, numpy=np.__version__))))
    np.savez_compressed(os.path.join(OUT, 'adaround_model.npz'), **data)
    print('layers', names, 'draws', len(drawn), 'act', list(zip(data['act_names'], data['act_delta'])))
    for n in names:
        print(n, 'ups', int((data[f'{n}_alpha'] >= 0).sum()), 'of', data[f'{n}_alpha'].size)
    print('out change', float(np.abs(data['out_after'] - data['out_before']).max()))


if __name__ == '__main__':
    main()
