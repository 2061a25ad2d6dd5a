"""(f2) fused residual-add -> quant -> LayerNorm -> quant.  Two statements:

* EXACT: against the oracle chain whose LayerNorm sums its fp32 statistics in the kernel's order (oracle/ln_sum.py ->
  oracle/tq_ln_oracle.c, one correctly rounded operation per step): every output equal, bit for bit.
* TOLERANCE against the reference's own op: torch.nn.functional.layer_norm leaves the summation order to the backend
  (torch CPU, torch GPU and this kernel all differ), so outputs that sit within round-off of a rounding boundary of the
  LAST quantizer may land one grid step apart: >= 99.9 % of elements identical to the torch-CPU chain, all others exactly
  one step away, and the un-quantized LayerNorm output within 1e-5 relative."""
import numpy as np
import pytest
import torch

from oracle import tq_oracle as O

pytestmark = pytest.mark.gpu


def _oracle_chain(a, r, q1, q2, w, b, eps, q3, kernel_order=False):
    def q(v, p):
        if p is None:
            return v
        delta, zf, n_bits, sym, sgn = p
        return O.fake_quant(v, delta, zf, n_bits, sym, sgn)[1]
    u = q(q(a.float(), q1) + r.float(), q2)
    if kernel_order:
        from oracle.ln_sum import layer_norm_kernel_order
        v = layer_norm_kernel_order(u, w, b, eps, a.dtype)
    else:
        v = torch.nn.functional.layer_norm(u, (u.shape[-1],), w, b, eps)
    return q(v, q3), v


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('d', [768, 3072, 512, 128, 1024, 256, 2048])
def test_fused_chain_equals_kernel_order_oracle(dtype, d):
    """Bit for bit against the chain whose LayerNorm statistics follow the kernel's summation order (every (lanes per row,
    vectors per lane) instantiation of launch_res_ln is hit by one of these widths)."""
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(d + 7)
    rows = 515                                               # not a multiple of the rows a block handles
    a = (torch.randn(rows, d, generator=g) * 2).to(dtype)
    r = (torch.randn(rows, d, generator=g) * 1.5).to(dtype)
    r[:, 5] *= 12
    w = 1 + 0.1 * torch.randn(d, generator=g)
    b = 0.05 * torch.randn(d, generator=g)
    d1, z1 = O.asym_params_from_range(-7.0, 7.5, 8)
    d2, z2 = O.asym_params_from_range(-20.0, 22.0, 8)
    d3, z3 = O.asym_params_from_range(-6.0, 11.0, 8)
    dev = lambda t: t.cuda()
    for use in ((1, 1, 1), (0, 1, 1), (1, 0, 0), (0, 0, 0), (1, 1, 0)):
        q1 = (d1, z1, 8, False, False) if use[0] else None
        q2 = (d2, z2, 8, False, False) if use[1] else None
        q3 = (d3, z3, 8, False, False) if use[2] else None
        ref, _ = _oracle_chain(a, r, q1, q2, w, b, 1e-12, q3, kernel_order=True)
        k = lambda q: None if q is None else (dev(q[0]), dev(q[1]), None, 8, False, False, 1e-8)
        out = be.residual_layernorm_quant(dev(a), dev(r), k(q1), k(q2), dev(w), dev(b), 1e-12, k(q3),
                                          want_idx=q3 is not None and dtype == torch.float32)
        y = (out[0] if isinstance(out, tuple) else out).cpu()
        assert y.dtype == dtype
        assert torch.equal(y, ref.to(dtype)), (use, float((y.float() - ref.to(dtype).float()).abs().max()),
                                               float((y != ref.to(dtype)).float().mean()))
        if isinstance(out, tuple):
            ref_idx = O.fake_quant(_oracle_chain(a, r, q1, q2, w, b, 1e-12, None, kernel_order=True)[0], d3, z3, 8, False)[0]
            assert torch.equal(out[1].cpu().float() + 128, ref_idx), use


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('d', [768, 3072, 512, 128])
def test_fused_chain_vs_oracle(dtype, d):
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(d)
    rows = 1024
    a = (torch.randn(rows, d, generator=g) * 2).to(dtype)
    r = (torch.randn(rows, d, generator=g) * 1.5).to(dtype)
    r[:, 5] *= 12
    w = 1 + 0.1 * torch.randn(d, generator=g)
    b = 0.05 * torch.randn(d, generator=g)
    d1, z1 = O.asym_params_from_range(-7.0, 7.5, 8)
    d2, z2 = O.asym_params_from_range(-20.0, 22.0, 8)
    d3, z3 = O.asym_params_from_range(-6.0, 11.0, 8)
    dev = lambda t: t.cuda()
    for use in ((1, 1, 1), (0, 1, 1), (1, 0, 0), (0, 0, 0)):
        q1 = (d1, z1, 8, False, False) if use[0] else None
        q2 = (d2, z2, 8, False, False) if use[1] else None
        q3 = (d3, z3, 8, False, False) if use[2] else None
        ref, ref_ln = _oracle_chain(a, r, q1, q2, w, b, 1e-12, q3)
        k = lambda q: None if q is None else (dev(q[0]), dev(q[1]), None, 8, False, False, 1e-8)
        y = be.residual_layernorm_quant(dev(a), dev(r), k(q1), k(q2), dev(w), dev(b), 1e-12, k(q3)).cpu()
        assert y.dtype == dtype
        ref_s = ref.to(dtype).float()
        diff = (y.float() - ref_s).abs()
        if q3 is None:
            assert torch.allclose(y.float(), ref_s, rtol=2e-5 if dtype == torch.float32 else 1e-2, atol=1e-5)
        else:
            step = float(d3)
            same = (diff == 0).float().mean().item()
            assert same >= 0.999, (use, same)
            assert float(diff.max()) <= step * 1.01 + (0 if dtype == torch.float32 else 0.1), (use, float(diff.max()))


def test_fused_block_in_bert_harness_matches_layered():
    """Whole BERT-base forward with the fused tails == layered forward up to single-step flips."""
    from tests.test_bert_e2e import _build, _fixture, _calibrate_and_run
    from tests.harness_bert import QResidualBlock
    z = _fixture()
    model, _ = _build('cuda')
    ids = torch.from_numpy(z['input_ids']).cuda()
    layered = _calibrate_and_run(model, ids)
    QResidualBlock.fuse = True
    try:
        with torch.no_grad():
            fused = model(ids)
    finally:
        QResidualBlock.fuse = False
    # single-step flips inside the 24 fused tails propagate like GEMM round-off does (see
    # test_bert_base_w8a8_gpu): most logits are bit-identical, the rest move by a few steps
    span = float(layered.max() - layered.min())
    assert float((fused - layered).abs().max()) <= 0.10 * span
    # (7-8 of the 16 logits on this random-init model; which ones depends on the order the fused kernel sums the
    # LayerNorm statistics in, so this is a sanity floor, not a parity bar -- the kernel-level tests above are)
    assert float(((fused - layered).abs() == 0).float().mean()) >= 0.25


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('d', [512, 128])
def test_nonorm_tail_is_bit_exact(dtype, d):
    """MobileBERT tail (NoNorm has no statistics, so nothing depends on a summation order): the fused kernel
    must equal the oracle chain bit for bit, indices included."""
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(d + 1)
    rows = 2048
    a = (torch.randn(rows, d, generator=g) * 2).to(dtype)
    r = (torch.randn(rows, d, generator=g) * 1.5).to(dtype)
    w = 1 + 0.3 * torch.randn(d, generator=g)
    b = 0.2 * torch.randn(d, generator=g)
    p1, p2, p3 = (O.asym_params_from_range(lo, hi, 8) for lo, hi in ((-7.0, 7.5), (-9.0, 11.0), (-12.0, 14.0)))
    k = lambda q: None if q is None else (q[0].cuda(), q[1].cuda(), None, 8, False, False, 1e-8)

    def q(v, p):
        return v if p is None else O.fake_quant(v, p[0], p[1], 8, False)[1]
    for use in ((1, 1, 1), (0, 1, 1), (1, 0, 0)):
        q1, q2, q3 = (p if u else None for p, u in zip((p1, p2, p3), use))
        u_ = q(q(a.float(), q1) + r.float(), q2)
        v_ = u_ * w + b
        ref = q(v_, q3).to(dtype)
        out = be.residual_layernorm_quant(a.cuda(), r.cuda(), k(q1), k(q2), w.cuda(), b.cuda(), None, k(q3),
                                          want_idx=q3 is not None)
        y = (out[0] if q3 is not None else out).cpu()
        assert y.dtype == dtype and torch.equal(y, ref), use
        if q3 is not None:
            ref_idx = O.fake_quant(v_, q3[0], q3[1], 8, False)[0]
            assert torch.equal(out[1].cpu().float() + 128, ref_idx)


def test_nonorm_block_module_level():
    """quantization.fused.residual_layernorm_quant with a QuantNoNorm == the layered MobileBERT modules."""
    from torch import nn
    from quantization.autoquant_utils import QuantNoNorm, quantize_model
    from quantization.base_quantized_classes import QuantizedActivation
    from quantization.fused import residual_layernorm_quant
    from quantization.quantizers import QMethods
    torch.manual_seed(0)
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8)

    class NoNorm(nn.Module):
        def __init__(self, d):
            super().__init__()
            self.weight = nn.Parameter(1 + 0.2 * torch.randn(d))
            self.bias = nn.Parameter(0.1 * torch.randn(d))
    dense = quantize_model(nn.Linear(128, 512), **qp).cuda().eval()
    resq = QuantizedActivation(**qp).cuda().eval()
    nn_ = QuantNoNorm(NoNorm(512), **qp).cuda().eval()
    x = torch.randn(4, 64, 128, device='cuda')
    res = torch.randn(4, 64, 512, device='cuda')
    with torch.no_grad():
        for m in (dense, resq, nn_):
            m.quantized()
        layered0 = nn_(resq(dense(x) + res))               # estimating: sets every range
        for m in (dense, resq, nn_):
            m.fix_ranges()
        layered = nn_(resq(dense(x) + res))
        fused = residual_layernorm_quant(dense, resq, nn_, x, res)
    assert torch.equal(fused, layered)


@pytest.mark.parametrize('dtype,d', [(torch.float32, d) for d in (64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 3072)] +
                         [(torch.bfloat16, d) for d in (128, 256, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144)])
def test_every_instantiated_row_length(dtype, d):
    """One case per (lanes-per-row, vectors-per-lane) instantiation of the tail kernel, LayerNorm and NoNorm."""
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(d)
    rows = 96
    a = (torch.randn(rows, d, generator=g) * 2).to(dtype)
    r = (torch.randn(rows, d, generator=g)).to(dtype)
    w = 1 + 0.1 * torch.randn(d, generator=g)
    b = 0.05 * torch.randn(d, generator=g)
    p1, p2, p3 = (O.asym_params_from_range(lo, hi, 8) for lo, hi in ((-7.0, 7.5), (-9.0, 10.0), (-6.0, 8.0)))
    k = lambda q: (q[0].cuda(), q[1].cuda(), None, 8, False, False, 1e-8)
    for eps in (1e-12, None):
        q1 = (p1[0], p1[1], 8, False, False)
        q2 = (p2[0], p2[1], 8, False, False)
        q3 = (p3[0], p3[1], 8, False, False)
        if eps is None:
            u = O.fake_quant(O.fake_quant(a.float(), p1[0], p1[1], 8, False)[1] + r.float(), p2[0], p2[1], 8, False)[1]
            ref = O.fake_quant(u * w + b, p3[0], p3[1], 8, False)[1].to(dtype)
        else:
            ref = _oracle_chain(a, r, q1, q2, w, b, eps, q3)[0].to(dtype)
        y = be.residual_layernorm_quant(a.cuda(), r.cuda(), k(p1), k(p2), w.cuda(), b.cuda(), eps, k(p3)).cpu()
        diff = (y.float() - ref.float()).abs()
        if eps is None:
            assert torch.equal(y, ref), (str(dtype), d)
        else:
            assert float((diff == 0).float().mean()) >= 0.998 and float(diff.max()) <= float(p3[0]) * 1.01 + (0.1 if dtype != torch.float32 else 0)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_fused_tails_special_values(dtype):
    """NaN / +-Inf / -0.0 inputs through the fused tails follow the reference chain: a NaN anywhere in a LayerNorm
    row makes the whole row NaN (the statistics are NaN), in a NoNorm row only its own element; +-Inf clamps to the
    grid ends like torch.clamp; outputs are never -0.0 where the reference produces +0.0."""
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(9)
    rows, d = 64, 768
    a = (torch.randn(rows, d, generator=g) * 2).to(dtype)
    r = (torch.randn(rows, d, generator=g) * 1.5).to(dtype)
    a[3, 17] = float('nan')
    r[7, 700] = float('nan')
    a[9, 5] = float('inf')
    r[11, 6] = -float('inf')
    a[13, :] = -0.0
    r[13, :] = -0.0
    w = 1 + 0.1 * torch.randn(d, generator=g)
    b = 0.05 * torch.randn(d, generator=g)
    p1, p2, p3 = (O.asym_params_from_range(lo, hi, 8) for lo, hi in ((-7.0, 7.5), (-9.0, 11.0), (-6.0, 11.0)))
    k = lambda q: (q[0].cuda(), q[1].cuda(), None, 8, False, False, 1e-8)
    fq = lambda v, p: O.fake_quant(v, p[0], p[1], 8, False)[1]
    u = fq(fq(a.float(), p1) + r.float(), p2)
    for eps in (1e-12, None):
        if eps is None:
            ref = fq(u * w + b, p3)
        else:
            ref = fq(torch.nn.functional.layer_norm(u, (d,), w, b, eps), p3)
        ref = ref.to(dtype).float()
        y = be.residual_layernorm_quant(a.cuda(), r.cuda(), k(p1), k(p2), w.cuda(), b.cuda(), eps, k(p3)).cpu().float()
        assert torch.equal(torch.isnan(y), torch.isnan(ref)), eps
        if eps is None:
            assert int(torch.isnan(y).sum()) == 2
            ok = ~torch.isnan(ref)
            assert torch.equal(y[ok].view(torch.int32), ref[ok].view(torch.int32))       # bit patterns incl. sign of zero
        else:
            assert torch.isnan(y[3]).all() and torch.isnan(y[7]).all() and int(torch.isnan(y).sum()) == 2 * d
            ok = ~torch.isnan(ref)
            same = (y[ok] == ref[ok]).float().mean().item()
            assert same >= 0.999
            assert not (y[ok].view(torch.int32) == -2 ** 31).any()                      # no -0.0


@pytest.mark.parametrize('d', [768, 1024, 128, 512])
def test_fused_embeddings_equal_kernel_order_oracle(d):
    """BERT's embedding block as ONE launch (tq_embeddings_layernorm_quant_fwd: word + token-type look-up, Q1, + position
    look-up, Q2, LayerNorm, Q3; reference models/quantized_bert.py:75-111) against the oracle chain with the LayerNorm
    statistics in the kernel's order: every output and int8 index equal, bit for bit -- incl. repeated ids and the quantizer subsets; out-of-range
    ids give NaN rows and a deferred IndexError."""
    from oracle.ln_sum import layer_norm_kernel_order
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(d)
    B, T, V_, P_ = 5, 37, 211, 64
    word = torch.randn(V_, d, generator=g) * 0.8
    typ = torch.randn(2, d, generator=g) * 0.3
    pos = torch.randn(P_, d, generator=g) * 0.5
    word[:, 3] *= 9
    ids = torch.randint(0, V_, (B, T), generator=g)
    ids[0, :4] = ids[0, 0]
    tok = torch.randint(0, 2, (B, T), generator=g)
    pid = torch.arange(T).unsqueeze(0).expand(B, T).contiguous()
    w = 1 + 0.1 * torch.randn(d, generator=g)
    b = 0.05 * torch.randn(d, generator=g)
    p1, p2, p3 = (O.asym_params_from_range(lo, hi, 8) for lo, hi in ((-8.0, 9.0), (-9.0, 10.0), (-5.0, 7.0)))

    def q(v, p):
        return v if p is None else O.fake_quant(v, p[0], p[1], 8, False)[1]
    k = lambda p: None if p is None else (p[0].cuda(), p[1].cuda(), None, 8, False, False, 1e-8)
    for use in ((1, 1, 1), (0, 1, 1), (1, 1, 0), (0, 0, 0)):
        q1, q2, q3 = (p if u else None for p, u in zip((p1, p2, p3), use))
        u2 = q(q(word[ids] + typ[tok], q1) + pos[pid], q2).reshape(-1, d)
        v = layer_norm_kernel_order(u2, w, b, 1e-12, torch.float32)
        ref = q(v, q3)
        out = be.embeddings_layernorm_quant(word.cuda(), ids.cuda(), typ.cuda(), tok.cuda(), pos.cuda(), pid.cuda(), k(q1), k(q2),
                                            w.cuda(), b.cuda(), 1e-12, k(q3), want_idx=q3 is not None)
        y = (out[0] if q3 is not None else out).cpu()
        assert torch.equal(y, ref), (use, float((y - ref).abs().max()))
        if q3 is not None:
            assert torch.equal(out[1].cpu().float() + 128, O.fake_quant(v, q3[0], q3[1], 8, False)[0]), use
    # ids outside their table (torch's CPU F.embedding raises IndexError): no out-of-bounds read, exactly the rows that
    # carry one are NaN, every other row is untouched, and the error surfaces -- once -- at the next synchronisation point
    be.raise_deferred(sync=True)
    bad = ids.clone()
    bad[1, 1], bad[2, 2] = V_ + 5, -3
    y_ok = be.embeddings_layernorm_quant(word.cuda(), ids.cuda(), typ.cuda(), tok.cuda(), pos.cuda(), pid.cuda(), k(p1), k(p2),
                                         w.cuda(), b.cuda(), 1e-12, k(p3)).cpu().view(*ids.shape, -1)
    y_bad = be.embeddings_layernorm_quant(word.cuda(), bad.cuda(), typ.cuda(), tok.cuda(), pos.cuda(), pid.cuda(), k(p1), k(p2),
                                          w.cuda(), b.cuda(), 1e-12, k(p3)).cpu().view(*ids.shape, -1)
    hit = torch.zeros(ids.shape, dtype=torch.bool)
    hit[1, 1] = hit[2, 2] = True
    assert torch.isnan(y_bad[hit]).all() and torch.equal(y_bad[~hit], y_ok[~hit])
    with pytest.raises(IndexError):
        be.raise_deferred(sync=True)
    be.raise_deferred(sync=True)              # reported once
    bad_pos = pid.clone()
    bad_pos.view(-1)[0] = pos.shape[0]          # the position / token-type tables are checked too
    be.embeddings_layernorm_quant(word.cuda(), ids.cuda(), typ.cuda(), tok.cuda(), pos.cuda(), bad_pos.cuda(), k(p1), k(p2),
                                  w.cuda(), b.cuda(), 1e-12, k(p3))
    with pytest.raises(IndexError):           # without an explicit check: at the next call
        torch.cuda.synchronize()
        be.embeddings_layernorm_quant(word.cuda(), ids.cuda(), typ.cuda(), tok.cuda(), pos.cuda(), pid.cuda(), k(p1), k(p2),
                                      w.cuda(), b.cuda(), 1e-12, k(p3))


@pytest.mark.layered_route
def test_fused_embeddings_in_bert_harness():
    """QEmbeddings.fuse: the block of the harness model as one launch -- used (one backend call per forward), >= 99.9 %
    of the outputs identical to the layered modules (torch's LayerNorm sums in another order), the rest one step away; the
    layered path runs while ranges are being estimated and under a forward hook on any of the modules involved."""
    from tests.test_bert_e2e import _build, _fixture, _calibrate_and_run
    from harness.bert import QEmbeddings
    from quantization import _hip
    z = _fixture()
    model, _ = _build('cuda')
    ids = torch.from_numpy(z['input_ids']).cuda()
    _calibrate_and_run(model, ids)
    emb = model.embeddings
    be = _hip.backend()
    calls = []
    orig = be.embeddings_layernorm_quant
    be.embeddings_layernorm_quant = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            layered = emb(ids)
            QEmbeddings.fuse = True
            fused = emb(ids)
            assert len(calls) == 1
            step = float(emb.LayerNorm.activation_quantizer.quantizer._delta)
            diff = (fused - layered).abs()
            assert float((diff == 0).float().mean()) >= 0.999 and float(diff.max()) <= step * 1.01
            h = emb.sum_pos_embd_act_quantizer.register_forward_hook(lambda m, i, o: None)
            try:
                assert torch.equal(emb(ids), layered) and len(calls) == 1          # an observer: layered modules
            finally:
                h.remove()
            model.estimate_ranges()
            emb(ids)
            assert len(calls) == 1                                                 # estimating: layered modules
    finally:
        QEmbeddings.fuse = False
        be.__dict__.pop('embeddings_layernorm_quant', None)
