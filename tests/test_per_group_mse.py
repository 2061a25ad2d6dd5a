"""Per-embedding-group MSE search (explicit extension; BASELINE configs[2]).
Default behaviour stays the reference's degenerate per-tensor search (covered by the
'mse-peg-degenerate' golden trace); with `per_group_search = True` each group of embedding dims
gets its own thresholds.  Oracle: the reference's MSE_Estimator(per_channel=True) on the
[n_groups, -1] view (vectors captured in tests/golden/estimators.npz, keys pg_*)."""
import numpy as np
import pytest
import torch

from tests._cases import t

CASES = (('asym4', 'asymmetric_uniform', 4, 'batches', dict(num_candidates=12)),
         ('sym8', 'symmetric_uniform', 8, 'batches', dict(num_candidates=50)),
         ('onesided6', 'asymmetric_uniform', 6, 'pos', dict(num_candidates=30)))


def _run(device, golden_estimators):
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    z, _ = golden_estimators
    for tag, method, n_bits, inp, ip in CASES:
        q = QMethods[method].cls(n_bits=n_bits)
        est = RangeEstimators.MSE.cls(quantizer=q, axis=2, n_groups=4, **ip)
        est.per_group_search = True
        xs = [t(b) for b in z['batches']]
        if inp == 'pos':
            xs = [x.abs() for x in xs]
        for b, x in enumerate(xs):
            mn, mx = est(x.to(device))
            assert mn.shape == (24,) and mx.shape == (24,)
            assert torch.equal(mn.cpu(), t(z[f'pg_{tag}_xmin'][b])), (tag, b)
            assert torch.equal(mx.cpu(), t(z[f'pg_{tag}_xmax'][b])), (tag, b)
        ref = z[f'pg_{tag}_loss']
        got = est.loss_array
        fin = np.isfinite(ref)
        assert got.shape == ref.shape
        assert np.allclose(got[fin], ref[fin], rtol=2e-5, atol=1e-6), tag
        if method == 'asymmetric_uniform':
            # the per-group ranges drive a per-embedding quantizer without complaint
            q.axis = 2
            q.set_quant_range(mn, mx)
            y = q(xs[-1].to(device))
            assert y.shape == xs[-1].shape


def test_per_group_mse_cpu(golden_estimators):
    from quantization import _hip
    from tests._oracle_backend import OracleBackend
    prev = _hip.set_backend(OracleBackend())
    try:
        _run('cpu', golden_estimators)
    finally:
        _hip.set_backend(prev)


@pytest.mark.gpu
def test_per_group_mse_gpu(golden_estimators):
    _run('cuda', golden_estimators)


@pytest.mark.gpu
def test_grouped_losses_match_transposed_rows_at_bert_shape():
    """[8,128,768], 6 groups, bf16 + fp32: the in-place grouped kernel == the contiguous-row kernel
    on the explicitly transposed tensor."""
    from quantization import _hip
    from quantization.range_estimators import candidate_params
    be = _hip.backend()
    g = torch.Generator().manual_seed(8)
    for dt in (torch.float32, torch.bfloat16):
        x = (torch.randn(8, 128, 768, generator=g) * 2).to(dt).cuda()
        cand = be.candidate_table(candidate_params(-np.linspace(0.5, 9, 40), np.linspace(0.5, 9, 40), 8, False), x.device)
        a = be.mse_candidates_grouped(x, 6, cand, be.zeros_f64((6, 40), x.device))
        xt = x.transpose(0, 2).contiguous().view(768, -1).view(6, -1)
        b = be.mse_candidates(xt, 6, cand, be.zeros_f64((6, 40), x.device))
        assert torch.allclose(a, b, rtol=2e-6, atol=0), (a - b).abs().max()   # fp32 lane partials, different order
