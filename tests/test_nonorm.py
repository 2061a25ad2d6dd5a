"""QuantNoNorm (MobileBERT W4A4 widths): reference-generated 3-batch trace
(tests/golden/nonorm.npz), incl. the shared-quantizer quirk q9; fixed-range output goes through the
fused affine + fake-quant kernel on the GPU."""
import os

import numpy as np
import pytest
import torch
from torch import nn

from tests.conftest import GOLDEN


class _Affine(nn.Module):
    def __init__(self, w, b):
        super().__init__()
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(b)


def _run(device, z, k):
    from quantization.quantizers import QMethods
    from quantization.autoquant_utils import QuantNoNorm
    d, w_bits, a_bits = (int(v) for v in z[f'n{k}_cfg'])
    org = _Affine(torch.from_numpy(z[f'n{k}_w']).clone(), torch.from_numpy(z[f'n{k}_b']).clone())
    m = QuantNoNorm(org, method=QMethods.symmetric_uniform, n_bits=w_bits,
                    act_method=QMethods.asymmetric_uniform, n_bits_act=a_bits).to(device)
    m.quantized()
    xs = [torch.from_numpy(x).to(device) for x in z[f'n{k}_x']]
    with torch.no_grad():
        ys = [m(x) for x in xs]
        m.weight_quantizer.fix_ranges()
        m.activation_quantizer.fix_ranges()
        y_fixed = m(xs[0])
    return m, ys, y_fixed


def _check(device, exact=True):
    z = np.load(os.path.join(GOLDEN, 'nonorm.npz'))
    for k in range(2):
        m, ys, y_fixed = _run(device, z, k)
        assert torch.equal(m.weight_quantizer.quantizer._delta.cpu(),
                           torch.from_numpy(z[f'n{k}_w_delta']))           # range of the BIAS (q9)
        assert torch.equal(m.activation_quantizer.quantizer._delta.cpu(), torch.from_numpy(z[f'n{k}_a_delta']))
        assert torch.equal(m.activation_quantizer.quantizer._zero_float.cpu(), torch.from_numpy(z[f'n{k}_a_zf']))
        for i, y in enumerate(ys):
            assert torch.equal(y.cpu(), torch.from_numpy(z[f'n{k}_y'][i])), (k, i)
        assert torch.equal(y_fixed.cpu(), torch.from_numpy(z[f'n{k}_y_fixed'])), k


def test_quant_nonorm_cpu():
    from quantization import _hip
    from tests._oracle_backend import OracleBackend
    prev = _hip.set_backend(OracleBackend())
    try:
        _check('cpu')
    finally:
        _hip.set_backend(prev)


@pytest.mark.gpu
def test_quant_nonorm_gpu():
    _check('cuda')


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_fused_affine_quant_matches_unfused_at_mobilebert_shapes(dtype):
    """[B*T, 512] and [B*T, 128] activations (MobileBERT hidden / bottleneck), W4A4."""
    from oracle import tq_oracle as O
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(1)
    for rows, d in ((1024, 512), (1024, 128), (4096 * 64, 512)):
        x = torch.randn(rows, d, generator=g).to(dtype)
        w = 1 + 0.2 * torch.randn(d, generator=g)
        b = 0.1 * torch.randn(d, generator=g)
        delta, zf = O.asym_params_from_range(-3.0, 4.0, 4)
        r = x.float() * w + b
        _, ref = O.fake_quant(r, delta, zf, 4, False)
        y = be.affine_fake_quant(x.cuda(), w.cuda(), b.cuda(), delta.cuda(), zf.cuda(), None, 4, False,
                                 False, 1e-8)
        assert torch.equal(y.cpu(), ref.to(dtype)), (rows, d, dtype)
        # optional int8(index - 128) output for a following integer Linear: same launch, same y
        y2, yi = be.affine_fake_quant(x.cuda(), w.cuda(), b.cuda(), delta.cuda(), zf.cuda(), None, 4, False,
                                      False, 1e-8, want_idx=True)
        ref_i, _ = O.fake_quant(r, delta, zf, 4, False)
        assert torch.equal(y2.cpu(), ref.to(dtype)) and torch.equal(yi.cpu().int() + 128, ref_i.int())
