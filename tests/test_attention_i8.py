"""(f3) tq_attention_i8_fwd: the attention core on int8 grid indices.  Reference = the layered chain on
the dequantised tensors evaluated in float64 for the two GEMMs (i.e. the exact values the reference's
fp32 GEMMs approximate) with the oracle's quantizers and torch's fp32 softmax in between.  exp() / row
sums differ in the last ulp between implementations, so a probability (or a context value) within
round-off of a rounding boundary may land one grid step away: bars are >= 99.5 % of the context outputs
bit-identical, none further than 2 steps (a flipped probability index moves a context value by at most
s_p |v|, which can cross one more boundary)."""
import math

import pytest
import torch

from oracle import tq_oracle as O

pytestmark = pytest.mark.gpu


def _params(lo, hi):
    d, z = O.asym_params_from_range(lo, hi, 8)
    return d, z


def _dq(idx8, p):
    d, z = p
    zp = torch.clamp(torch.round(z), 0, 255)
    return (d * ((idx8.double() + 128) - zp.double()))


def _reference(qi, ki, vi, H, mask, pq, pk, pv, ps, pp, pc):
    B, T, D = qi.shape
    dh = D // H
    split = lambda x: x.view(B, T, H, dh).permute(0, 2, 1, 3)
    Q, K, V = split(_dq(qi, pq)), split(_dq(ki, pk)), split(_dq(vi, pv))
    S = torch.matmul(Q, K.transpose(-1, -2)).float()
    if ps is not None:
        S = O.fake_quant(S, ps[0], ps[1], 8, False)[1]
    S = S / math.sqrt(dh)
    if mask is not None:
        S = S + mask.view(B, 1, 1, T)
    P = O.fake_quant(torch.softmax(S, dim=-1), pp[0], pp[1], 8, False)[1]
    C = torch.matmul(P.double(), V).float().permute(0, 2, 1, 3).reshape(B, T, D)
    if pc is not None:
        return O.fake_quant(C, pc[0], pc[1], 8, False)
    return None, C


@pytest.mark.parametrize('T', [64, 128, 192, 256, 384, 512])
def test_attention_i8_vs_float64_reference(T):
    from quantization import _hip
    be = _hip.backend()
    B, H = 2, 3
    D = H * 64
    g = torch.Generator().manual_seed(T)
    qi, ki, vi = (torch.randint(-128, 128, (B, T, D), generator=g, dtype=torch.int8) for _ in range(3))
    keep = (torch.rand(B, T, generator=g) > 0.25).float()
    keep[:, 0] = 1
    mask = (1 - keep) * -10000.0
    pq, pk, pv = _params(-3.0, 2.5), _params(-2.0, 3.0), _params(-1.5, 1.0)
    ps, pp, pc = _params(-60.0, 70.0), _params(0.0, 0.6), _params(-1.2, 0.9)
    dev = lambda t: None if t is None else t.cuda()
    k7 = lambda p: None if p is None else (dev(p[0]), dev(p[1]), None, 8, False, False, 1e-8)
    for use_s, use_c, use_m in ((1, 1, 1), (0, 1, 1), (1, 1, 0), (1, 0, 1)):
        ref_idx, ref = _reference(qi, ki, vi, H, mask if use_m else None, pq, pk, pv,
                                  ps if use_s else None, pp, pc if use_c else None)
        out = be.attention_i8(dev(qi), dev(ki), dev(vi), H, dev(mask) if use_m else None, 8.0, k7(pq), k7(pk), k7(pv),
                              k7(ps) if use_s else None, k7(pp), k7(pc) if use_c else None, want_idx=bool(use_c))
        ctx = (out[0] if use_c else out).cpu()
        diff = (ctx - ref).abs()
        if use_c:
            step = float(pc[0])
            assert float((diff == 0).float().mean()) >= 0.995, (T, use_s, use_m, float((diff == 0).float().mean()))
            assert float(diff.max()) <= 2.01 * step
            idx = out[1].cpu().float() + 128
            assert torch.equal(idx * 0 + ctx, ctx) and float((idx - ref_idx).abs().max()) <= 2
            # the emitted indices are the indices of the emitted values
            zp = torch.clamp(torch.round(pc[1]), 0, 255)
            assert torch.equal(pc[0] * (idx - zp), ctx)
        else:
            # un-quantized context: a flipped probability index shifts a value by <= s_p * max|v|
            # (a few probability indices per row may flip by one step; the more keys, the more candidates)
            tol = float(pp[0]) * float(_dq(vi, pv).abs().max()) * 4
            assert float(diff.max()) <= tol, (float(diff.max()), tol)
            assert float((diff <= 1e-5 * ref.abs() + 1e-6).float().mean()) >= 0.9


def test_attention_i8_rejects_unsupported():
    from quantization import _hip
    be = _hip.backend()
    z = lambda *s: torch.zeros(*s, dtype=torch.int8, device='cuda')
    p = lambda: (torch.tensor(0.1).cuda(), torch.tensor(3.0).cuda(), None, 8, False, False, 1e-8)
    with pytest.raises(_hip.TQError):       # T = 96
        be.attention_i8(z(1, 96, 128), z(1, 96, 128), z(1, 96, 128), 2, None, 8.0, p(), p(), p(), None, p(), None)
    with pytest.raises(_hip.TQError):       # head_dim = 16
        be.attention_i8(z(1, 64, 128), z(1, 64, 128), z(1, 64, 128), 8, None, 4.0, p(), p(), p(), None, p(), None)
    sym = (torch.tensor(0.1).cuda(), None, torch.tensor(True).cuda(), 8, True, False, 1e-8)
    with pytest.raises(_hip.TQError):       # symmetric probabilities grid
        be.attention_i8(z(1, 64, 128), z(1, 64, 128), z(1, 64, 128), 2, None, 8.0, p(), p(), p(), None, sym, None)


def test_bert_forward_with_integer_attention_matches_layered():
    from tests.test_bert_e2e import _build, _fixture, _calibrate_and_run
    from tests.harness_bert import QResidualBlock, QSelfAttention
    from quantization import options
    z = _fixture()
    model, _ = _build('cuda')
    ids = torch.from_numpy(z['input_ids']).cuda()
    layered = _calibrate_and_run(model, ids)
    QResidualBlock.fuse = QSelfAttention.fuse = True
    options.INT8_LINEAR = True
    try:
        from quantization import _hip
        calls, gcalls = [], []
        orig, gorig = _hip.HipBackend.attention_i8, _hip.HipBackend.linear_i8_grouped
        _hip.HipBackend.attention_i8 = lambda self, *a, **k: (calls.append(1), orig(self, *a, **k))[1]
        _hip.HipBackend.linear_i8_grouped = lambda self, *a, **k: (gcalls.append(1), gorig(self, *a, **k))[1]
        with torch.no_grad():
            fused = model(ids)
        _hip.HipBackend.attention_i8, _hip.HipBackend.linear_i8_grouped = orig, gorig
    finally:
        QResidualBlock.fuse = QSelfAttention.fuse = False
        options.INT8_LINEAR = False
    assert len(calls) == 12, 'the integer attention kernel must serve all 12 layers'
    assert len(gcalls) == 12, 'query / key / value must run as one grouped GEMM per layer'
    span = float(layered.max() - layered.min())
    assert float((fused - layered).abs().max()) <= 0.10 * span


def test_stacked_qkv_projection_equals_three_linears():
    """quantized_self_attention (grouped QKV GEMM, index-only output, strided attention reads) must be
    bit-identical to three separate integer Linears followed by quantized_attention."""
    from tests.test_bert_e2e import _build, _fixture, _calibrate_and_run
    from quantization import options
    from quantization.fused import quantized_attention, quantized_self_attention
    z = _fixture()
    model, _ = _build('cuda')
    ids = torch.from_numpy(z['input_ids']).cuda()
    _calibrate_and_run(model, ids)
    options.INT8_LINEAR = True
    try:
        with torch.no_grad():
            h = model.embeddings(ids)                      # tagged with its quantizer + int8 indices
            from quantization import provenance
            assert provenance.indices_of(h) is not None
            A = model.layers[0].attention_self
            mask = torch.zeros(ids.shape[0], 1, 1, ids.shape[1], device='cuda')
            mask[1, ..., 100:] = -10000.0
            args = (mask, A.heads, A.attn_scores_act_quantizer, A.attn_probs_act_quantizer, A.context_act_quantizer)
            one = quantized_self_attention(h, A.query, A.key, A.value, *args)
            sep = quantized_attention(A.query(h), A.key(h), A.value(h), *args)
    finally:
        options.INT8_LINEAR = False
    assert one is not None and sep is not None
    assert torch.equal(one, sep) and torch.equal(provenance.indices_of(one), provenance.indices_of(sep))


def test_mobilebert_grouped_query_key_equals_three_linears():
    """MobileBERT's query and key Linears read the SAME bottlenecked tensor, the value Linear the layer input:
    quantized_self_attention((q_in, k_in, v_in), ...) runs Q | K as one grouped index-only launch, V as another, and the
    core reads V with a row stride of its own (tq_attention_i8_strided_fwd) -- bit-identical to three integer Linears
    followed by quantized_attention."""
    from tests.test_mobilebert_e2e import _fixture as mb_fixture
    from harness.mobilebert import build_mobilebert
    from quantization import _hip, options, provenance
    from quantization.fused import quantized_attention, quantized_self_attention
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from utils.utils import pass_data_for_range_estimation
    z = mb_fixture()
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
              weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)
    model, _ = build_mobilebert(seed=1000, num_layers=2, **qp)
    model = model.cuda().eval()
    ids = torch.from_numpy(z['input_ids']).cuda()
    calls = []
    be = _hip.backend()
    orig = be.linear_i8_grouped
    with torch.no_grad():
        pass_data_for_range_estimation([(ids,)], model, act_quant=True, weight_quant=True, max_num_batches=1)
        model.fix_ranges()
        options.INT8_LINEAR = True
        be.linear_i8_grouped = lambda *a, **k: (calls.append(len(a[8])), orig(*a, **k))[1]
        try:
            L = model.layers[0]
            h = model.embeddings(ids)
            assert provenance.indices_of(h) is not None
            shared = L.bottleneck_attention(h)
            assert provenance.indices_of(shared) is not None
            A = L.attention_self
            mask = torch.zeros(ids.shape[0], 1, 1, ids.shape[1], device='cuda')
            mask[2, ..., 90:] = -10000.0
            args = (mask, A.heads, A.attn_scores_act_quantizer, A.attn_probs_act_quantizer, A.attn_output_act_quantizer)
            one = quantized_self_attention((shared, shared, h), A.query, A.key, A.value, *args)
            sep = quantized_attention(A.query(shared), A.key(shared), A.value(h), *args)
            # three different tensor objects: three launches, same result
            shared2 = L.bottleneck_attention(h)
            calls_before = len(calls)
            three = quantized_self_attention((shared, shared2, h), A.query, A.key, A.value, *args)
        finally:
            options.INT8_LINEAR = False
            be.__dict__.pop('linear_i8_grouped', None)
    assert one is not None and sep is not None and three is not None
    assert calls[:2] == [2, 1] and calls[calls_before:] == [1, 1, 1]
    assert torch.equal(one, sep) and torch.equal(provenance.indices_of(one), provenance.indices_of(sep))
    assert torch.equal(three, sep)


@pytest.mark.parametrize('n_bits', [4, 6])
def test_attention_i8_low_bit_grids(n_bits):
    """W4A4-style configurations: Q / K / V / probabilities / context on 4- and 6-bit grids (indices are
    still carried as int8(index - 128))."""
    from quantization import _hip
    be = _hip.backend()
    B, H, T = 2, 2, 128
    D = H * 64
    top = 2 ** n_bits - 1
    g = torch.Generator().manual_seed(n_bits)
    qi, ki, vi = ((torch.randint(0, top + 1, (B, T, D), generator=g) - 128).to(torch.int8) for _ in range(3))
    P = lambda lo, hi: O.asym_params_from_range(lo, hi, n_bits)
    pq, pk, pv, ps, pp, pc = P(-3.0, 2.5), P(-2.0, 3.0), P(-1.5, 1.0), P(-40.0, 50.0), P(0.0, 0.3), P(-0.6, 0.5)

    def dq(idx8, p):
        zp = torch.clamp(torch.round(p[1]), 0, top)
        return p[0] * ((idx8.double() + 128) - zp.double())
    split = lambda x: x.view(B, T, H, 64).permute(0, 2, 1, 3)
    S = torch.matmul(split(dq(qi, pq)), split(dq(ki, pk)).transpose(-1, -2)).float()
    S = O.fake_quant(S, ps[0], ps[1], n_bits, False)[1] / 8.0
    Pm = O.fake_quant(torch.softmax(S, dim=-1), pp[0], pp[1], n_bits, False)[1]
    C = torch.matmul(Pm.double(), split(dq(vi, pv))).float().permute(0, 2, 1, 3).reshape(B, T, D)
    ref_idx, ref = O.fake_quant(C, pc[0], pc[1], n_bits, False)
    k7 = lambda p: (p[0].cuda(), p[1].cuda(), None, n_bits, False, False, 1e-8)
    ctx, idx = be.attention_i8(qi.cuda(), ki.cuda(), vi.cuda(), H, None, 8.0, k7(pq), k7(pk), k7(pv), k7(ps), k7(pp), k7(pc),
                               want_idx=True)
    diff = (ctx.cpu() - ref).abs()
    assert float((diff == 0).float().mean()) >= 0.99
    assert float(diff.max()) <= 2.01 * float(pc[0])
    assert float((idx.cpu().float() + 128 - ref_idx).abs().max()) <= 2


@pytest.mark.parametrize('T', [128, 384])
def test_attention_i8_head_dim_32(T):
    """MobileBERT geometry: 4 heads of 32 dims (half of the MFMA K step is zero padding)."""
    from quantization import _hip
    be = _hip.backend()
    B, H, dh = 2, 4, 32
    D = H * dh
    g = torch.Generator().manual_seed(T + 7)
    qi, ki, vi = (torch.randint(-128, 128, (B, T, D), generator=g, dtype=torch.int8) for _ in range(3))
    mask = ((torch.rand(B, T, generator=g) > 0.8).float()) * -10000.0
    mask[:, 0] = 0
    pq, pk, pv = _params(-3.0, 2.5), _params(-2.0, 3.0), _params(-1.5, 1.0)
    ps, pp, pc = _params(-30.0, 35.0), _params(0.0, 0.6), _params(-1.2, 0.9)
    ref_idx, ref = _reference(qi, ki, vi, H, mask, pq, pk, pv, ps, pp, pc)
    k7 = lambda p: (p[0].cuda(), p[1].cuda(), None, 8, False, False, 1e-8)
    ctx, idx = be.attention_i8(qi.cuda(), ki.cuda(), vi.cuda(), H, mask.cuda(), math.sqrt(dh), k7(pq), k7(pk), k7(pv), k7(ps),
                               k7(pp), k7(pc), want_idx=True)
    diff = (ctx.cpu() - ref).abs()
    assert float((diff == 0).float().mean()) >= 0.995
    assert float(diff.max()) <= 2.01 * float(pc[0])
    assert float((idx.cpu().float() + 128 - ref_idx).abs().max()) <= 2


def test_stacked_qkv_with_per_channel_weights():
    """Grouped QKV GEMM with per-output-channel weight scales (the stacked per-row scale vector mixes the three
    layers' scales) == three separate integer Linears, bit for bit."""
    from torch import nn
    from quantization import options
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_classes import QuantizedActivation
    from quantization.fused import quantized_attention, quantized_self_attention
    from quantization.quantizers import QMethods
    torch.manual_seed(3)
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
              per_channel_weights=True)
    D, H, B, T = 256, 4, 4, 64
    mods = [quantize_model(nn.Linear(D, D), **qp).cuda().eval() for _ in range(3)]
    qin, qs, qpr, qc = (QuantizedActivation(**qp).cuda().eval() for _ in range(4))
    x = torch.randn(B, T, D, device='cuda')
    with torch.no_grad():
        for m in mods + [qin, qs, qpr, qc]:
            m.quantized()
        h = qin(x)
        q, k, v = (m(h) for m in mods)
        sc = qs(torch.matmul(q.view(B, T, H, 64).permute(0, 2, 1, 3), k.view(B, T, H, 64).permute(0, 2, 3, 1)))
        pr = qpr(torch.softmax(sc / 8.0, dim=-1))
        qc(torch.matmul(pr, v.view(B, T, H, 64).permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(B, T, D))
        for m in mods + [qin, qs, qpr, qc]:
            m.fix_ranges()
        assert mods[0].weight_quantizer.quantizer._delta.numel() == D      # really per channel
        options.INT8_LINEAR = True
        try:
            h = qin(x)
            one = quantized_self_attention(h, *mods, None, H, qs, qpr, qc)
            sep = quantized_attention(mods[0](h), mods[1](h), mods[2](h), None, H, qs, qpr, qc)
        finally:
            options.INT8_LINEAR = False
    assert one is not None and sep is not None and torch.equal(one, sep)


@pytest.mark.parametrize('fill', [float('-inf'), torch.finfo(torch.float32).min])
def test_attention_i8_with_infinite_masks(fill):
    """Newer HF versions mask with -inf / finfo.min instead of -10000: masked keys must get probability index z_p
    exactly, un-masked rows must be unaffected."""
    from quantization import _hip
    be = _hip.backend()
    B, H, T = 2, 2, 128
    D = H * 64
    g = torch.Generator().manual_seed(11)
    qi, ki, vi = (torch.randint(-128, 128, (B, T, D), generator=g, dtype=torch.int8) for _ in range(3))
    mask = torch.zeros(B, T)
    mask[0, 40:] = fill
    mask[1, ::3] = fill
    mask[1, 0] = 0
    pq, pk, pv = _params(-3.0, 2.5), _params(-2.0, 3.0), _params(-1.5, 1.0)
    ps, pp, pc = _params(-60.0, 70.0), _params(0.0, 0.6), _params(-1.2, 0.9)
    ref_idx, ref = _reference(qi, ki, vi, H, mask, pq, pk, pv, ps, pp, pc)
    k7 = lambda p: (p[0].cuda(), p[1].cuda(), None, 8, False, False, 1e-8)
    ctx, idx = be.attention_i8(qi.cuda(), ki.cuda(), vi.cuda(), H, mask.cuda(), 8.0, k7(pq), k7(pk), k7(pv), k7(ps), k7(pp),
                               k7(pc), want_idx=True)
    assert torch.isfinite(ctx).all()
    diff = (ctx.cpu() - ref).abs()
    assert float((diff == 0).float().mean()) >= 0.995 and float(diff.max()) <= 2.01 * float(pc[0])


@pytest.mark.parametrize('T,dh,denom', [(64, 64, 8.0), (128, 64, 8.0), (128, 32, math.sqrt(32.0)), (256, 64, 3.0), (192, 32, 8.0)])
def test_branch_free_chain_is_bit_identical_to_the_guarded_one(T, dh, denom, monkeypatch):
    """T <= 256 runs the exact branch-free quantizer / quotient chain (QF + Markstein); TQ_ATTN_FAST=0 keeps the
    guarded-reciprocal one.  Same context values and indices bit for bit, with and without the scores quantizer, with
    -inf masks, and NaN for a fully masked batch row in both."""
    from quantization import _hip
    be = _hip.backend()
    B, H = 3, 2
    g = torch.Generator().manual_seed(11 * T + dh)
    qi, ki, vi = (torch.randint(-128, 128, (B, T, H * dh), generator=g, dtype=torch.int8).cuda() for _ in range(3))
    mask = torch.zeros(B, T)
    mask[1, T // 3:] = -float('inf')
    mask[2, :] = -float('inf')
    pq, pk, pv = _params(-3.0, 2.5), _params(-2.0, 3.0), _params(-1.5, 1.0)
    ps, pp, pc = _params(-60.0, 70.0), _params(0.0, 0.6), _params(-1.2, 0.9)
    k7 = lambda p: None if p is None else (p[0].cuda(), p[1].cuda(), None, 8, False, False, 1e-8)
    for use_s in (True, False):
        for m in (mask.cuda(), None):
            outs = []
            for fast in ('1', '0'):
                monkeypatch.setenv('TQ_ATTN_FAST', fast)
                ctx, idx = be.attention_i8(qi, ki, vi, H, m, denom, k7(pq), k7(pk), k7(pv), k7(ps) if use_s else None,
                                           k7(pp), k7(pc), want_idx=True)
                outs.append((ctx.cpu(), idx.cpu()))
            (c1, i1), (c0, i0) = outs
            assert torch.equal(torch.isnan(c1), torch.isnan(c0))
            ok = ~torch.isnan(c1)
            assert torch.equal(c1[ok].view(torch.int32), c0[ok].view(torch.int32))
            assert torch.equal(i1[ok], i0[ok])
            if m is not None:
                assert torch.isnan(c1[2]).all() and not torch.isnan(c1[:2]).any()


@pytest.mark.parametrize('T,dh', [(128, 64), (256, 64), (128, 32), (512, 64)])
def test_key_split_workgroups_match_the_two_wave_form(T, dh, monkeypatch):
    """Small grids split the keys of a query tile over two waves (row max / row sum / integer partial sums through
    LDS).  Only the order of the float row sum differs from the two-wave form: probability indices may move by one
    step on a rounding tie, so >= 99.9 % of the context values are bit-identical and the rest one context step away;
    the branch-free and the guarded element chains stay bit-identical within each form."""
    from quantization import _hip
    be = _hip.backend()
    B, H = 2, 3
    g = torch.Generator().manual_seed(5 * T + dh)
    qi, ki, vi = (torch.randint(-128, 128, (B, T, H * dh), generator=g, dtype=torch.int8).cuda() for _ in range(3))
    mask = torch.zeros(B, T)
    mask[1, T // 2:] = -10000.0
    pq, pk, pv = _params(-3.0, 2.5), _params(-2.0, 3.0), _params(-1.5, 1.0)
    ps, pp, pc = _params(-60.0, 70.0), _params(0.0, 0.6), _params(-1.2, 0.9)
    k7 = lambda p: (p[0].cuda(), p[1].cuda(), None, 8, False, False, 1e-8)
    res = {}
    for split in ('1', '0'):
        for fast in ('1', '0'):
            monkeypatch.setenv('TQ_ATTN_SPLIT', split)
            monkeypatch.setenv('TQ_ATTN_FAST', fast)
            ctx, idx = be.attention_i8(qi, ki, vi, H, mask.cuda(), math.sqrt(dh), k7(pq), k7(pk), k7(pv), k7(ps), k7(pp),
                                       k7(pc), want_idx=True)
            res[split, fast] = (ctx.cpu(), idx.cpu())
    for split in ('1', '0'):
        assert torch.equal(res[split, '1'][0], res[split, '0'][0]) and torch.equal(res[split, '1'][1], res[split, '0'][1])
    a, b = res['1', '1'], res['0', '1']
    same = (a[0] == b[0]).float().mean()
    assert float(same) >= 0.999, float(same)
    assert int((a[1].int() - b[1].int()).abs().max()) <= 1


@pytest.mark.parametrize('T,dh', [(128, 64), (128, 32), (256, 64), (384, 32), (512, 64)])
def test_eight_query_waves_per_workgroup_are_bit_identical(T, dh, monkeypatch):
    """Large grids run the one-wave-per-row form with EIGHT query waves per workgroup (TQ_ATTN_QW=8: the V tile is fetched
    and transposed once per 128 queries instead of once per 32): same per-wave code on the same data, so values, indices
    and NaN rows are those of the two-wave workgroups, bit for bit -- with masks, without the scores quantizer, branch-free
    and guarded chains."""
    from quantization import _hip
    be = _hip.backend()
    B, H = 3, 2
    g = torch.Generator().manual_seed(13 * T + dh)
    qi, ki, vi = (torch.randint(-128, 128, (B, T, H * dh), generator=g, dtype=torch.int8).cuda() for _ in range(3))
    mask = torch.zeros(B, T)
    mask[1, T // 3:] = -10000.0
    mask[2, :] = -float('inf')
    pq, pk, pv = _params(-3.0, 2.5), _params(-2.0, 3.0), _params(-1.5, 1.0)
    ps, pp, pc = _params(-60.0, 70.0), _params(0.0, 0.6), _params(-1.2, 0.9)
    k7 = lambda p: None if p is None else (p[0].cuda(), p[1].cuda(), None, 8, False, False, 1e-8)
    monkeypatch.setenv('TQ_ATTN_SPLIT', '0')
    for use_s in (True, False):
        for fast in ('1', '0'):
            monkeypatch.setenv('TQ_ATTN_FAST', fast)
            outs = []
            for qw in ('2', '8'):
                monkeypatch.setenv('TQ_ATTN_QW', qw)
                ctx, idx = be.attention_i8(qi, ki, vi, H, mask.cuda(), math.sqrt(dh), k7(pq), k7(pk), k7(pv),
                                           k7(ps) if use_s else None, k7(pp), k7(pc), want_idx=True)
                outs.append((ctx.cpu(), idx.cpu()))
            (c1, i1), (c2, i2) = outs
            assert torch.isnan(c1[2]).all() and not torch.isnan(c1[:2]).any()
            assert torch.equal(torch.isnan(c1), torch.isnan(c2))
            ok = ~torch.isnan(c1)
            assert torch.equal(c1[ok].view(torch.int32), c2[ok].view(torch.int32)) and torch.equal(i1[ok], i2[ok])


@pytest.mark.parametrize('zq', [0.0, 1.0, 128.0, 255.0])
@pytest.mark.parametrize('T,dh,split', [(128, 64, '1'), (128, 64, '0'), (256, 32, '0'), (64, 64, '0')])
def test_query_zero_point_extremes_equal_the_integer_oracle(zq, T, dh, split, monkeypatch):
    """Round 6: the zero-point correction c_q sum_d a'_k of the scores is an MFMA of the K tile against an operand whose
    bytes are all c_q = 128 - z_q; c_q = 128 (z_q = 0: a one-sided query grid) does not fit a byte and runs as two passes
    with 64.  Every z_q incl. the ends of the grid equals oracle/tq_int_oracle.c bit for bit."""
    from oracle import int_oracle as IO
    from quantization import _hip
    be = _hip.backend()
    B, H = 2, 2
    g = torch.Generator().manual_seed(int(zq) * 7 + T + dh)
    qi, ki, vi = (torch.randint(-128, 128, (B, T, H * dh), generator=g).to(torch.int8) for _ in range(3))
    mask = torch.zeros(B, T)
    mask[1, T - 9:] = -10000.0
    mk = lambda d, z, nb=8: (torch.tensor(d), torch.tensor(z), None, nb, False, False, 1e-8)
    q_q, q_k, q_v = mk(0.011, zq), mk(0.013, 131.0), mk(0.009, 128.0)
    q_s, q_p, q_c = mk(0.35, 128.0), mk(1.0 / 255, 0.0), mk(0.012, 125.0)
    f = lambda q: (float(q[0]), float(q[1]), None, q[3], False, False, q[6])
    dev = lambda q: (q[0].cuda(), q[1].cuda(), None, q[3], False, False, q[6])
    monkeypatch.setenv('TQ_ATTN_SPLIT', split)
    denom = math.sqrt(dh)
    ctx, ci = be.attention_i8(qi.cuda(), ki.cuda(), vi.cuda(), H, mask.cuda(), denom, dev(q_q), dev(q_k), dev(q_v), dev(q_s),
                              dev(q_p), dev(q_c), want_idx=True)
    ref, ri = IO.attention_i8(qi, ki, vi, H, mask, denom, f(q_q), f(q_k), f(q_v), f(q_s), f(q_p), f(q_c))
    assert torch.equal(ci.cpu(), ri) and torch.equal(ctx.cpu(), ref)
