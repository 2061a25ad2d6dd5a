"""tq_mse_candidates_ordered: candidate losses in the reference's own fp32 summation order.

The reference's MSE_Estimator.loss_fx (/root/reference/quantization/range_estimators.py:248-256) returns
`torch.sum(torch.sum(err.view(len(data), -1), dim=1))` computed by ATen's CPU cascade-sum kernel; the HIP
kernel must return the SAME BITS (scipy's golden-section iterates and the grid argmin depend on them).
Checked against oracle/aten_sum.py (numpy restatement, pinned against torch.sum by the CPU suite) and
against torch.sum on this box's CPU.
"""
import os

import numpy as np
import pytest
import torch

from oracle import aten_sum as A

pytestmark = pytest.mark.gpu
DEV = 'cuda'
# ATen splits a reduction whose OUTPUT is a single element over threads once the input has >= 32768 elements
# (TensorIterator::parallel_reduce), so the reference's value is thread-count dependent there (len(data) == 1
# activations, >= 32768 rows); the kernel implements the single-thread order, compared like-for-like.
torch.set_num_threads(1)


def _err(x, c):
    s, zp, lo, hi = (float(v) for v in c)
    s = torch.tensor(s, dtype=torch.float32)
    xi = torch.clamp(torch.round(x / s) + zp, lo, hi)
    return (x - s * (xi - zp)) ** 2


def _table(rs, n, x, n_bits=8, sym=False):
    hi = float(2 ** n_bits - 1)
    scales = (float(x.abs().max()) * rs.uniform(0.05, 1.2, n) / hi).astype(np.float32)
    if sym:
        return np.stack([scales, np.zeros(n, np.float32), np.full(n, -2.0 ** (n_bits - 1), np.float32),
                         np.full(n, 2.0 ** (n_bits - 1) - 1, np.float32)], 1).astype(np.float32)
    return np.stack([scales, rs.randint(0, int(hi) + 1, n).astype(np.float32), np.zeros(n, np.float32),
                     np.full(n, hi, np.float32)], 1).astype(np.float32)


SHAPES = [(768, 768), (8, 128, 768), (3072,), (768,), (5, 7), (100, 5), (1, 40003), (3, 100003), (2, 3, 4),
          (1,), (7,), (8,), (33, 9), (4, 16384 + 33), (2, 131072 + 4096 + 77), (64, 3072), (30522, 8)]


@pytest.mark.parametrize('ktop', [0, 2, 3, 4])
@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_ordered_losses_bit_exact(shape, ktop):
    from quantization import _hip
    be = _hip.backend()
    rs = np.random.RandomState(len(shape) * 1000 + shape[0] + ktop)
    g = torch.Generator().manual_seed(int(np.prod(shape)) % 9973 + ktop)
    x = torch.randn(*shape, generator=g) * 1.7
    x.view(-1)[::97] *= 9.0
    n_cand = 1 if (shape[0] + ktop) % 3 == 0 else (11 if np.prod(shape) > 500000 else 21)
    tab = _table(rs, n_cand, x, sym=bool(shape[0] % 2))
    cand = torch.from_numpy(tab)
    os.environ['TQ_ORD_KTOP'] = str(ktop)
    try:
        for per_row in (False, True):
            rows = shape[0]
            loss = torch.full((rows if per_row else 1, n_cand), 0.5, dtype=torch.float64, device=DEV)
            _, f32 = be.mse_candidates_ordered(x.to(DEV), cand.to(DEV), loss, per_row=per_row, want_f32=True)
            got = f32.cpu().numpy()
            for ci in range(n_cand):
                e = _err(x, tab[ci])
                ref = A.loss_sum(e.numpy().reshape(rows, -1), per_channel_loss=per_row)
                tor = torch.sum(e.view(rows, -1), dim=1)
                tor = tor.numpy() if per_row else torch.sum(tor).numpy()
                assert np.array_equal(got[:, ci], np.atleast_1d(ref)), (shape, ktop, per_row, ci)
                if True:
                    assert np.array_equal(np.atleast_1d(ref), np.atleast_1d(tor)), (shape, per_row, ci)
            # fp64 accumulation cell: previous content + (double) fp32 loss
            assert np.array_equal(loss.cpu().numpy(), 0.5 + got.astype(np.float64))
    finally:
        os.environ.pop('TQ_ORD_KTOP', None)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_ordered_losses_low_precision_storage(dtype):
    """bf16 / fp16 storage is widened to fp32 in registers; the sum order is that of the fp32 tensor."""
    from quantization import _hip
    be = _hip.backend()
    rs = np.random.RandomState(3)
    x = (torch.randn(16, 64, 96, generator=torch.Generator().manual_seed(5)) * 2).to(dtype)
    xf = x.float()
    tab = _table(rs, 9, xf)
    _, f32 = be.mse_candidates_ordered(x.to(DEV), torch.from_numpy(tab).to(DEV), None, want_f32=True)
    for ci in range(9):
        ref = A.loss_sum(_err(xf, tab[ci]).numpy().reshape(16, -1))
        assert f32[0, ci].item() == float(ref)


def test_ordered_matches_unordered_to_rounding():
    """The fp64-accumulating kernel (per-group extension) and the ordered kernel see the same element errors."""
    from quantization import _hip
    be = _hip.backend()
    rs = np.random.RandomState(4)
    x = torch.randn(8, 128, 768, generator=torch.Generator().manual_seed(6))
    tab = _table(rs, 100, x)
    cand = torch.from_numpy(tab).to(DEV)
    a = be.mse_candidates(x.to(DEV), 1, cand, be.zeros_f64((1, 100), DEV)).cpu().numpy()
    b, _ = be.mse_candidates_ordered(x.to(DEV), cand, be.zeros_f64((1, 100), DEV))
    assert np.allclose(a, b.cpu().numpy(), rtol=5e-6, atol=0)
