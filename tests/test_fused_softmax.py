"""(f2) fused attention probabilities: Q_probs(softmax(Q_scores(s) / sqrt(d) + mask)).  exp() and the row
sum differ in the last ulp between implementations, so a probability within round-off of a rounding
boundary of Q_probs may land one grid step away: >= 99.9 % of elements bit-identical to the CPU oracle
chain, the rest exactly one step off; without Q_probs the probabilities agree to 1e-5 relative (+1e-7)."""
import math

import pytest
import torch

from oracle import tq_oracle as O

pytestmark = pytest.mark.gpu


def _oracle_chain(s, mask, denom, q1, q2):
    def q(v, p):
        if p is None:
            return v
        return O.fake_quant(v, p[0], p[1], 8, False, False)[1]
    t = q(s, q1) / denom
    if mask is not None:
        t = t + mask
    return q(torch.softmax(t, dim=-1), q2)


@pytest.mark.parametrize('T', [32, 64, 128, 256, 512, 1024])
def test_fused_softmax_vs_oracle(T):
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(T)
    B, H, Tq = 3, 4, 40
    s = torch.randn(B, H, Tq, T, generator=g) * 24
    keep = (torch.rand(B, T, generator=g) > 0.2).float()
    keep[:, 0] = 1
    mask = ((1 - keep) * -10000.0).reshape(B, 1, 1, T)
    p1 = O.asym_params_from_range(-70.0, 80.0, 8)
    p2 = O.asym_params_from_range(0.0, 1.0, 8)
    k = lambda p: None if p is None else (p[0].cuda(), p[1].cuda(), None, 8, False, False, 1e-8)
    for use1, use2, use_mask in ((1, 1, 1), (0, 1, 1), (1, 0, 1), (1, 1, 0), (0, 0, 0)):
        q1, q2 = (p1 if use1 else None), (p2 if use2 else None)
        m = mask if use_mask else None
        ref = _oracle_chain(s, m, 8.0, q1, q2)
        y = be.scores_softmax_quant(s.cuda(), None if m is None else m.reshape(B, T).cuda(), H * Tq, 8.0,
                                    k(q1), k(q2)).cpu()
        diff = (y - ref).abs()
        if q2 is None:
            assert torch.allclose(y, ref, rtol=1e-5, atol=1e-7), float(diff.max())
        else:
            assert float((diff == 0).float().mean()) >= 0.999
            assert float(diff.max()) <= float(p2[0]) * 1.01


def test_fused_softmax_rejects_bad_shapes():
    from quantization import _hip
    be = _hip.backend()
    s = torch.randn(2, 2, 4, 48, device='cuda')
    with pytest.raises(_hip.TQError):
        be.scores_softmax_quant(s, None, 8, 8.0, None, None)


def test_module_level_entry_matches_layered_modules():
    """quantization.fused.scores_softmax_quant on fixed QuantizedActivations == the layered calls; while
    the quantizers are still estimating it must run (and update) the layered path."""
    from quantization.base_quantized_classes import QuantizedActivation
    from quantization.fused import scores_softmax_quant
    from quantization.quantizers import QMethods
    g = torch.Generator().manual_seed(5)
    s = (torch.randn(2, 12, 128, 128, generator=g) * 20).cuda()
    mask = torch.zeros(2, 1, 1, 128).cuda()
    mask[1, ..., 100:] = -10000.0
    qa, qb = (QuantizedActivation(act_method=QMethods.asymmetric_uniform, n_bits_act=8).cuda() for _ in range(2))
    for m in (qa, qb):
        m.quantized_acts()
    denom = math.sqrt(64)
    est = scores_softmax_quant(qa, qb, s, mask, denom)          # estimating -> layered path, ranges set
    assert qa.activation_quantizer.quantizer.is_initialized and qb.activation_quantizer.quantizer.is_initialized
    for m in (qa, qb):
        m.fix_ranges()
    layered = qb(torch.softmax(qa(s) / denom + mask, dim=-1))
    fused = scores_softmax_quant(qa, qb, s, mask, denom)
    step = float(qb.activation_quantizer.quantizer.delta)
    d = (fused - layered).abs()
    assert float((d == 0).float().mean()) >= 0.999 and float(d.max()) <= step * 1.01
    assert torch.equal(est, layered)


def test_fused_attention_in_bert_harness_matches_layered():
    from tests.test_bert_e2e import _build, _fixture, _calibrate_and_run
    from tests.harness_bert import QSelfAttention
    z = _fixture()
    model, _ = _build('cuda')
    ids = torch.from_numpy(z['input_ids']).cuda()
    layered = _calibrate_and_run(model, ids)
    QSelfAttention.fuse = True
    try:
        with torch.no_grad():
            fused = model(ids)
    finally:
        QSelfAttention.fuse = False
    span = float(layered.max() - layered.min())
    assert float((fused - layered).abs().max()) <= 0.10 * span


@pytest.mark.parametrize('T,denom', [(32, math.sqrt(32.0)), (128, 8.0), (512, 3.0)])
def test_branch_free_softmax_is_bit_identical_to_the_division_path(T, denom, monkeypatch):
    """The exact-quotient body (Markstein x / denom and e / sum, QF quantizers) against the same kernel's IEEE-division
    body: every output bit, including NaN rows, +-inf scores, -inf / finfo.min masks and a fully masked row."""
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(7 * T)
    B, H, Tq = 4, 3, 33
    s = torch.randn(B, H, Tq, T, generator=g) * 30
    s[0, 0, 0, 3] = float('nan')
    s[0, 0, 1, 5] = float('inf')
    s[0, 0, 2, 7] = -float('inf')
    s[1, 1, 4] = 0.0
    mask = torch.zeros(B, T)
    mask[1, T // 2:] = -float('inf')
    mask[2, 1:] = torch.finfo(torch.float32).min
    mask[3, :] = -float('inf')                      # fully masked: NaN rows in the reference too
    p1 = O.asym_params_from_range(-70.0, 80.0, 8)
    p2 = O.asym_params_from_range(0.0, 1.0, 8)
    k = lambda p: (p[0].cuda(), p[1].cuda(), None, 8, False, False, 1e-8)
    outs = []
    for fast in ('1', '0'):
        monkeypatch.setenv('TQ_SM_FAST', fast)
        for m in (mask.cuda(), None):
            outs.append(be.scores_softmax_quant(s.cuda(), m, H * Tq, denom, k(p1), k(p2)).cpu())
    for a, b in ((outs[0], outs[2]), (outs[1], outs[3])):
        assert torch.equal(torch.isnan(a), torch.isnan(b))
        assert torch.equal(torch.nan_to_num(a, nan=-1.0).view(torch.int32), torch.nan_to_num(b, nan=-1.0).view(torch.int32))
    assert torch.isnan(outs[0][0, 0, 0]).all() and torch.isnan(outs[0][3]).all()
    assert not torch.isnan(outs[1][1:]).any()
