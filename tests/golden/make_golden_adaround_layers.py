#!/usr/bin/env python3
"""Golden vectors for AdaRound on the NON-Linear layer kinds the reference's `'all'` selection covers
(utils/adaround_utils.py:60-75: every QuantizedModule with a `weight`): the word-embedding table `[30522, 768]` and a
LayerNorm weight `[768]`, at BERT-base's real shapes, by IMPORTING the reference (build container only; the .npz is data).

    python tests/golden/make_golden_adaround_layers.py

The embedding table is 93.8 MB, far too big for a fixture: weights are REGENERATED on both sides from numpy's legacy
Mersenne-Twister stream (`np.random.RandomState(seed).standard_normal`, bit-stable across numpy versions and
platforms); the fixture holds the token ids, the configuration and the reference's results -- losses, grid, the learned
alpha on a sample of touched and untouched rows (full for the LayerNorm), per-row sums of the hard-rounded integer
weights and the total number of round-ups.
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
from quantization.adaround import apply_adaround_to_layer  # noqa: E402
from quantization.adaround.config import DEFAULT_ADAROUND_CONFIG  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
VOCAB, DIM, OUT_F = 30522, 768, 32
W_SEED, ITERS, LR, BS, N_SAMPLES, T = 9100, 40, 2e-2, 8, 32, 128


def weights():
    """Deterministic BERT-like parameters (initializer_range 0.02; LayerNorm weight around 1 with spread, as trained
    checkpoints have)."""
    rs = np.random.RandomState(W_SEED)
    emb = (rs.standard_normal((VOCAB, DIM)) * 0.02).astype(np.float32)
    ln_w = (1.0 + 0.25 * rs.standard_normal(DIM)).astype(np.float32)
    ln_b = (0.05 * rs.standard_normal(DIM)).astype(np.float32)
    fc_w = (rs.standard_normal((OUT_F, DIM)) * 0.02).astype(np.float32)
    fc_b = np.zeros(OUT_F, np.float32)
    return emb, ln_w, ln_b, fc_w, fc_b


class EmbNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.emb = nn.Embedding(VOCAB, DIM)
        self.ln = nn.LayerNorm(DIM, eps=1e-12)
        self.fc = nn.Linear(DIM, OUT_F)
        emb, ln_w, ln_b, fc_w, fc_b = weights()
        with torch.no_grad():
            self.emb.weight.copy_(torch.from_numpy(emb))
            self.ln.weight.copy_(torch.from_numpy(ln_w))
            self.ln.bias.copy_(torch.from_numpy(ln_b))
            self.fc.weight.copy_(torch.from_numpy(fc_w))
            self.fc.bias.copy_(torch.from_numpy(fc_b))

    def forward(self, ids):
        return self.fc(self.ln(self.emb(ids)))


class QEmbNet(QuantizedModel):
    def __init__(self, org, **qp):
        super().__init__()
        self.emb = quantize_model(org.emb, **qp)
        self.ln = quantize_model(org.ln, **qp)
        self.fc = quantize_model(org.fc, **qp)

    def forward(self, ids):
        return self.fc(self.ln(self.emb(ids)))


def main():
    torch.set_num_threads(8)
    data, meta = {}, []
    g = torch.Generator().manual_seed(9200)
    ids = torch.randint(1000, VOCAB, (N_SAMPLES, T), generator=g)
    ids[:, 0] = 101
    ids[:, -1] = 102
    data['ids'] = ids.numpy()
    org = EmbNet()
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=4, n_bits_act=8,
              weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)

    def fresh():
        m = QEmbNet(copy.deepcopy(org), **qp)
        m.eval()
        m.set_quant_state(weight_quant=True, act_quant=False)
        with torch.no_grad():
            m(ids[:BS])
        for mod in m.modules():
            if isinstance(mod, QuantizationManager) and mod.quantizer.is_initialized:
                mod.fix_ranges()
        return m

    for k, lname in enumerate(('emb', 'ln')):
        m = fresh()
        layer = getattr(m, lname)
        cfg = copy.deepcopy(DEFAULT_ADAROUND_CONFIG)
        cfg.iters, cfg.lr = ITERS, LR
        m.full_precision()
        layer.quantized_weights()
        seed = 9300 + k
        torch.manual_seed(seed)
        res = apply_adaround_to_layer(m, layer, ids, batch_size=BS, act_quant=False, adaround_config=cfg)
        wq = layer.weight_quantizer.quantizer
        with torch.no_grad():
            hard_idx = wq.to_integer_forward(layer.weight)
        alpha = wq.alpha.detach()
        data[f'l{k}_delta'] = wq._delta.detach().numpy().astype(np.float32)
        data[f'l{k}_losses'] = np.array([res.loss_soft_before, res.loss_hard_before, res.loss_soft_after,
                                         res.loss_hard_after], dtype=np.float64)
        data[f'l{k}_ups'] = np.array(int((alpha >= 0).sum()), dtype=np.int64)
        if lname == 'emb':
            touched = torch.unique(ids)[:48]
            untouched = torch.tensor([i for i in range(0, 1000, 21)][:48])
            rows = torch.cat([touched, untouched])
            data[f'l{k}_rows'] = rows.numpy()
            data[f'l{k}_alpha_rows'] = alpha[rows].numpy().astype(np.float32)
            data[f'l{k}_hard_rowsum'] = hard_idx.sum(1).to(torch.int64).numpy()
        else:
            data[f'l{k}_alpha'] = alpha.numpy().astype(np.float32)
            data[f'l{k}_hard_idx'] = hard_idx.numpy().astype(np.float32)
        meta.append(dict(k=k, layer=lname, iters=ITERS, lr=LR, bs=BS, seed=seed, n_bits=4))
        print(lname, 'losses', data[f'l{k}_losses'], 'ups', int(data[f'l{k}_ups']), flush=True)
    data['meta'] = np.array(json.dumps(dict(cases=meta, vocab=VOCAB, dim=DIM, out_features=OUT_F, w_seed=W_SEED,
                                            versions=dict(torch=torch.__version__, numpy=np.__version__))))
    np.savez_compressed(os.path.join(OUT, 'adaround_layers.npz'), **data)
    print('wrote adaround_layers.npz')


if __name__ == '__main__':
    main()
