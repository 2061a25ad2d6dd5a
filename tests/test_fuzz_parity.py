"""Seeded fuzz of the fake-quant / statistics kernels against the CPU oracle: random shapes (aligned, ragged,
odd row lengths -> vector, register, LDS-table, row and scalar kernels), layouts (per-tensor, last axis, inner
axis, per-channel), dtypes, bit widths, symmetric / asymmetric.  Indices and dequantised values bit-exact."""
import numpy as np
import pytest
import torch

from oracle import tq_oracle as O

pytestmark = pytest.mark.gpu


def _case(rs):
    layout = rs.choice(['tensor', 'last', 'inner', 'channel'])
    dtype = [torch.float32, torch.bfloat16, torch.float16][rs.randint(3)]
    nd = rs.randint(1, 4)
    if layout == 'tensor':
        shape = tuple(int(rs.choice([1, 2, 3, 5, 7, 8, 16, 33, 64, 100, 257])) for _ in range(nd))
        axis, per_channel = None, False
    elif layout == 'last':
        d = int(rs.choice([4, 8, 12, 24, 40, 64, 96, 100, 128, 136, 520, 768, 1024, 1160, 2056]))
        shape = tuple(int(rs.choice([1, 2, 3, 5, 9, 17, 64])) for _ in range(rs.randint(1, 3))) + (d,)
        axis, per_channel = len(shape) - 1, False
    elif layout == 'inner':
        shape = (int(rs.choice([1, 2, 5])), int(rs.choice([3, 8, 12, 32])), int(rs.choice([1, 4, 7, 8, 24, 100])))
        axis, per_channel = 1, False
    else:
        shape = (int(rs.choice([1, 3, 8, 30])), int(rs.choice([1, 4, 7, 8, 96, 100])))
        axis, per_channel = None, True
    sym = bool(rs.randint(2)) and layout in ('tensor', 'channel')          # per-axis needs the asymmetric quantizer
    n_bits = int(rs.choice([2, 3, 4, 6, 8, 8, 8, 12, 16]))
    return layout, dtype, shape, axis, per_channel, sym, n_bits


@pytest.mark.parametrize('rows_flat', [None, '3'], ids=['by-size', 'flat-row-kernels'])
@pytest.mark.parametrize('seed', range(6))
def test_fake_quant_fuzz(seed, rows_flat, monkeypatch):
    # TQ_ROWS_FLAT=3: the inner-axis / per-channel layouts take the flat-tile kernels that only large launches reach by size
    # (fq_rows_tab; fq_rows_flat for rows of fewer than 4 vectors)
    if rows_flat is not None:
        monkeypatch.setenv('TQ_ROWS_FLAT', rows_flat)
    from quantization import _hip
    from quantization.quantizers import param_layout
    be = _hip.backend()
    rs = np.random.RandomState(1000 + seed)
    g = torch.Generator().manual_seed(seed)
    for _ in range(60):
        layout, dtype, shape, axis, per_channel, sym, n_bits = _case(rs)
        x = (torch.randn(*shape, generator=g) * float(rs.choice([0.1, 1.0, 30.0]))).to(dtype)
        if rs.rand() < 0.2 and x.numel() > 3:
            x.view(-1)[rs.randint(x.numel())] = float('inf')
        n_par = shape[axis] if axis is not None else (shape[0] if per_channel else 1)
        lo = -np.abs(rs.randn(n_par)).astype(np.float32) * 2 - 0.01
        hi = np.abs(rs.randn(n_par)).astype(np.float32) * 2 + 0.01
        if n_par == 1:
            lo, hi = lo[0], hi[0]
        if sym:
            delta, signed = O.sym_params_from_range(torch.as_tensor(lo), torch.as_tensor(hi), n_bits)
            zf, sgn = None, bool(signed)
        else:
            delta, zf = O.asym_params_from_range(torch.as_tensor(lo), torch.as_tensor(hi), n_bits)
            signed, sgn = None, False
        ref_idx, ref_y = O.fake_quant_lowp(x, delta, zf, n_bits, sym, sgn, axis=axis, per_channel=per_channel)
        n_params, inner = param_layout(x, int(delta.numel()), axis, per_channel, tuple(delta.shape))
        xd = x.cuda()
        if rs.rand() < 0.25 and x.numel() > 1:                       # unaligned base pointer -> scalar kernels
            buf = torch.empty(x.numel() + 1, dtype=dtype, device='cuda')
            xd = buf[1:].view(shape)
            xd.copy_(x)
        y, idx = be.fake_quant(xd, delta.reshape(-1).cuda(), None if zf is None else zf.reshape(-1).cuda(),
                               None if signed is None else signed.cuda(), n_bits, sym, False, 1e-8, n_params, inner,
                               idx_dtype=torch.int32)
        tag = (layout, str(dtype), shape, sym, n_bits)
        fin = torch.isfinite(ref_idx)
        assert torch.equal(idx.cpu().float()[fin], ref_idx[fin]), tag
        assert torch.equal(y.cpu()[fin], ref_y[fin]), tag


@pytest.mark.parametrize('seed', range(3))
def test_minmax_fuzz(seed):
    from quantization import _hip
    from quantization.quantizers import param_layout
    be = _hip.backend()
    rs = np.random.RandomState(2000 + seed)
    g = torch.Generator().manual_seed(100 + seed)
    for _ in range(60):
        layout, dtype, shape, axis, per_channel, _, _ = _case(rs)
        x = (torch.randn(*shape, generator=g) * 5).to(dtype)
        if layout == 'tensor':
            ref = O.minmax_tensor(x.float())
            n_params, inner = 1, 1
        elif per_channel:
            ref = O.minmax_axis(x.float(), 0)
            n_params, inner = shape[0], x.numel() // shape[0]
        else:
            ref = O.minmax_axis(x.float(), axis)
            n_params = shape[axis]
            inner = int(np.prod(shape[axis + 1:])) if axis + 1 < len(shape) else 1
        mn, mx = be.minmax(x.cuda(), n_params, inner)
        assert torch.equal(mn.cpu().reshape(-1), ref[0].reshape(-1)) and torch.equal(mx.cpu().reshape(-1), ref[1].reshape(-1)), \
            (layout, str(dtype), shape)


@pytest.mark.parametrize('fused', [True, False], ids=['fused-calibration', 'layered-calibration'])
@pytest.mark.parametrize('seed', range(3))
def test_calibration_fuzz(seed, fused):
    """Random estimator / layout / shape sequences through QuantizationManager (3 batches each): estimator state,
    quantizer parameters and the quantized last batch against the oracle's functional restatement."""
    from quantization import quantization_manager as qm
    from quantization.quantization_manager import QuantizationManager
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from utils.per_embd_quant_utils import set_act_quant_axis_and_groups
    rs = np.random.RandomState(3000 + seed)
    g = torch.Generator().manual_seed(200 + seed)
    prev = qm.FUSED_CALIBRATION
    qm.FUSED_CALIBRATION = fused
    try:
        for _ in range(40):
            est = str(rs.choice(['current_minmax', 'running_minmax', 'allminmax']))
            layout = str(rs.choice(['tensor', 'embd', 'groups', 'channel']))
            dtype = [torch.float32, torch.bfloat16][rs.randint(2)]
            n_bits = int(rs.choice([4, 8]))
            if layout == 'channel':
                shape, axis, per_channel, ng = (int(rs.choice([3, 8, 30])), int(rs.choice([7, 8, 96]))), None, True, None
            elif layout == 'tensor':
                shape, axis, per_channel, ng = tuple(int(rs.choice([2, 5, 16, 64, 100])) for _ in range(rs.randint(1, 4))), None, False, None
            else:
                d = int(rs.choice([24, 96, 120, 768]))
                shape, axis, per_channel = (int(rs.choice([1, 3, 8])), int(rs.choice([5, 16, 64])), d), 2, False
                ng = int(rs.choice([2, 3, 6])) if layout == 'groups' else None
            if est == 'allminmax' and layout in ('embd', 'groups'):
                est = 'running_minmax'                       # AllMinMax ignores axis / groups upstream (quirk q5)
            sym = per_channel and bool(rs.randint(2))
            mgr = QuantizationManager(qmethod=QMethods.symmetric_uniform if sym else QMethods.asymmetric_uniform,
                                      init=RangeEstimators[est], per_channel=per_channel, qparams=dict(n_bits=n_bits),
                                      init_params=dict(momentum=0.7) if est == 'running_minmax' else {})
            if axis is not None:
                set_act_quant_axis_and_groups(mgr, axis=axis, n_groups=ng)
            cur = (None, None)
            for b in range(3):
                x = (torch.randn(*shape, generator=g) * float(rs.choice([0.5, 4.0]))).to(dtype)
                y = mgr(x.cuda())
                new = O.batch_minmax(x.float(), axis=axis, n_groups=ng, per_channel=per_channel)
                if est == 'current_minmax':
                    cur = new
                elif est == 'running_minmax':
                    cur = O.running_update(cur[0], cur[1], new[0], new[1], 0.7)
                else:
                    cur = O.allminmax_update(cur[0], cur[1], new[0], new[1])
            tag = (est, layout, str(dtype), shape, ng, sym, n_bits)
            e, q = mgr.range_estimator, mgr.quantizer
            assert torch.equal(e.current_xmin.cpu().reshape(-1), cur[0].reshape(-1)), tag
            assert torch.equal(e.current_xmax.cpu().reshape(-1), cur[1].reshape(-1)), tag
            if sym:
                delta, signed = O.sym_params_from_range(cur[0], cur[1], n_bits)
                zf, sgn = None, bool(signed)
                assert bool(q._signed) == sgn, tag
            else:
                delta, zf = O.asym_params_from_range(cur[0], cur[1], n_bits)
                sgn = False
                assert torch.equal(q._zero_float.cpu().reshape(-1), zf.reshape(-1)), tag
            assert torch.equal(q._delta.cpu().reshape(-1), delta.reshape(-1)), tag
            _, ref_y = O.fake_quant_lowp(x, delta, zf, n_bits, sym, sgn, axis=axis, per_channel=per_channel)
            assert torch.equal(y.cpu(), ref_y), tag
    finally:
        qm.FUSED_CALIBRATION = prev


@pytest.mark.parametrize('seed', range(3))
def test_ste_backward_fuzz(seed):
    """dx of the STE backward kernel (g inside the clip range, 0 outside) against autograd through the oracle,
    bit-exact, over random shapes / dtypes / per-tensor and per-embedding parameters / (un)aligned pointers."""
    from quantization import _hip
    be = _hip.backend()
    rs = np.random.RandomState(4000 + seed)
    g = torch.Generator().manual_seed(300 + seed)
    for _ in range(40):
        dtype = [torch.float32, torch.bfloat16, torch.float16][rs.randint(3)]
        per_embd = bool(rs.randint(2))
        if per_embd:
            d = int(rs.choice([8, 24, 96, 100, 768]))
            shape = (int(rs.choice([1, 3, 17])), int(rs.choice([2, 16, 65])), d)
        else:
            shape = tuple(int(rs.choice([1, 3, 8, 33, 100, 257])) for _ in range(rs.randint(1, 4)))
        n_bits = int(rs.choice([2, 4, 8]))
        sym = (not per_embd) and bool(rs.randint(2))
        x = (torch.randn(*shape, generator=g) * 3).to(dtype)
        go = torch.randn(*shape, generator=g).to(dtype)
        n_par = shape[-1] if per_embd else 1
        lo = torch.as_tensor(-np.abs(rs.randn(n_par)).astype(np.float32) * 2 - 0.05)
        hi = torch.as_tensor(np.abs(rs.randn(n_par)).astype(np.float32) * 2 + 0.05)
        if n_par == 1:
            lo, hi = lo[0], hi[0]
        if sym:
            delta, signed = O.sym_params_from_range(lo, hi, n_bits)
            zf, sgn = None, bool(signed)
        else:
            delta, zf = O.asym_params_from_range(lo, hi, n_bits)
            signed, sgn = None, False
        _, ref_dx, _, _ = O.fake_quant_with_grads(x.float(), delta, zf, n_bits, sym, sgn, grad_out=go.float(),
                                                  axis=(len(shape) - 1) if per_embd else None)
        xd, gd = x.cuda(), go.cuda()
        if rs.rand() < 0.3 and x.numel() > 1:
            bx = torch.empty(x.numel() + 1, dtype=dtype, device='cuda')
            xd = bx[1:].view(shape)
            xd.copy_(x)
        gx, _, _ = be.fake_quant_bwd(xd, gd, delta.reshape(-1).cuda(), None if zf is None else zf.reshape(-1).cuda(),
                                     None if signed is None else signed.cuda(), n_bits, sym, False, 1e-8, n_par, 1)
        assert torch.equal(gx.cpu(), ref_dx.to(dtype)), (str(dtype), shape, per_embd, sym, n_bits)


@pytest.mark.parametrize('seed', range(2))
def test_mse_candidates_fuzz(seed):
    """Per-candidate squared-error sums (tq_mse_candidates: contiguous rows; tq_mse_candidates_grouped: column
    groups of a [tokens, d] tensor) against float64 sums of the oracle's per-element errors; rows / groups,
    ragged lengths, dtypes, candidate counts around the 128-candidate tile."""
    from quantization import _hip
    be = _hip.backend()
    rs = np.random.RandomState(5000 + seed)
    g = torch.Generator().manual_seed(400 + seed)
    for _ in range(25):
        dtype = [torch.float32, torch.bfloat16, torch.float16][rs.randint(3)]
        C = int(rs.choice([1, 7, 100, 128, 129, 300]))
        n_bits = int(rs.choice([4, 8]))
        top = 2.0 ** n_bits - 1
        scales = np.exp(rs.uniform(-5, 0, C)).astype(np.float32)
        zps = np.round(rs.uniform(0, top, C)).astype(np.float32)
        tab = np.stack([scales, zps, np.zeros(C, np.float32), np.full(C, top, np.float32)], 1)
        grouped = bool(rs.randint(2))
        if grouped:
            ng = int(rs.choice([2, 3, 6]))
            d = ng * int(rs.choice([4, 8, 20, 128]))
            x = (torch.randn(int(rs.choice([1, 7, 64, 300])), d, generator=g) * 2).to(dtype)
            rows_ref = [x.float().reshape(-1, ng, d // ng)[:, k, :].reshape(-1) for k in range(ng)]
            loss = be.zeros_f64((ng, C), 'cuda')
            be.mse_candidates_grouped(x.cuda(), ng, torch.from_numpy(tab).cuda(), loss)
        else:
            rows = int(rs.choice([1, 1, 3, 16]))
            L = int(rs.choice([1, 5, 64, 1000, 4097, 20000]))
            x = (torch.randn(rows, L, generator=g) * 2).to(dtype)
            rows_ref = [x.float()[r] for r in range(rows)]
            loss = be.zeros_f64((rows, C), 'cuda')
            be.mse_candidates(x.cuda(), rows, torch.from_numpy(tab).cuda(), loss)
        got = loss.cpu().numpy()
        for r, xr in enumerate(rows_ref):
            for c in range(0, C, max(1, C // 9)):
                s_, z_ = float(scales[c]), float(zps[c])
                xi = torch.clamp(torch.round(xr / s_) + z_, 0.0, top)
                ref = float(((xr - s_ * (xi - z_)).double() ** 2).sum())
                assert abs(got[r, c] - ref) <= 1e-5 * abs(ref) + 1e-12, (str(dtype), grouped, tuple(x.shape), C, r, c, got[r, c], ref)


@pytest.mark.parametrize('seed', range(2))
def test_adaround_kernels_fuzz(seed):
    """K10 (soft / hard AdaRound forward) and the alpha initialisation against the oracle over random weight
    shapes, per-tensor / per-channel symmetric and asymmetric grids and the three relaxations.  Hard rounding is
    bit-exact; the soft forward involves sigmoid / log (device libm vs SLEEF): 2e-5 of a grid step."""
    from quantization import _hip
    be = _hip.backend()
    rs = np.random.RandomState(6000 + seed)
    g = torch.Generator().manual_seed(500 + seed)
    modes = [('learned_sigmoid', _hip.ADA_SIGMOID), ('learned_hard_sigmoid', _hip.ADA_HARD_SIGMOID),
             ('sigmoid_temp_decay', _hip.ADA_SIGMOID_TEMP)]
    for _ in range(30):
        shape = (int(rs.choice([1, 3, 16, 64, 257])), int(rs.choice([1, 7, 64, 768])))
        per_channel = bool(rs.randint(2))
        sym = bool(rs.randint(2))
        n_bits = int(rs.choice([2, 4, 8]))
        mode, code = modes[rs.randint(3)]
        temp = float(rs.choice([2.0, 8.0, 20.0]))
        w = torch.randn(*shape, generator=g) * 0.1
        lo = w.amin(1) if per_channel else w.min()
        hi = w.amax(1) if per_channel else w.max()
        if sym:
            delta, signed = O.sym_params_from_range(lo, hi, n_bits)
            zf, sgn = None, bool(signed)
        else:
            delta, zf = O.asym_params_from_range(lo, hi, n_bits)
            signed, sgn = None, False
        n_par = shape[0] if per_channel else 1
        inner = shape[1] if per_channel else 1
        dev = lambda t: None if t is None else t.reshape(-1).cuda()
        qargs = (dev(delta), dev(zf), None if signed is None else signed.cuda(), n_bits, sym, False, 1e-8, n_par, inner)
        dB = delta.reshape(-1, 1) if per_channel else delta
        zB = None if zf is None else (zf.reshape(-1, 1) if per_channel else zf)
        scale = O.effective_scale(dB)
        # alpha initialisation
        a_ref = O.ada_alpha_init(w, scale, mode, temp)
        a_dev = be.adaround_init_alpha(w.cuda(), qargs, code, temp).cpu()
        fin = torch.isfinite(a_ref)
        assert torch.allclose(a_dev[fin], a_ref[fin], rtol=2e-4, atol=2e-4), (shape, mode, sym, n_bits)
        # forward with a shared alpha
        alpha = (torch.randn(*shape, generator=g) * 3)
        for soft in (True, False):
            _, ref = O.ada_fake_quant(w, alpha, dB, zB, n_bits, sym, sgn, mode, soft, temperature=temp)
            got = be.adaround_fwd(w.cuda(), alpha.cuda(), qargs, code, soft, temp).cpu()
            if soft:
                step = scale.expand_as(w) if torch.is_tensor(scale) else scale
                assert float(((got - ref).abs() / step).max()) <= 2e-5, (shape, mode, sym, n_bits, per_channel)
            else:
                assert torch.equal(got, ref), (shape, mode, sym, n_bits, per_channel)


@pytest.mark.parametrize('seed', range(4))
def test_integer_linear_epilogue_fuzz(seed):
    """Seeded fuzz of the integer Linear's fused epilogues: random tile-able shapes (LDS-staged 64 / 128 tiles and the
    LDS-free kernel), per-tensor / per-channel weight scales, bit widths, activation, output quantizer on / off, index
    output, NoNorm tail with / without residual and with any subset of its quantizers.  Reference: the SAME integer
    GEMM without epilogue extras followed by the stand-alone kernels (each bit-exact against the oracle elsewhere):
    quantizer and NoNorm-tail epilogues must agree bit for bit; GELU (branch-free erf vs libm) within 2.4e-7 before
    the quantizer and >= 99.95 % identical indices after it."""
    from quantization import _hip
    be = _hip.backend()
    rs = np.random.RandomState(4000 + seed)
    for case in range(10):
        M = int(rs.choice([32, 64, 96, 128, 256, 1024]))
        N = int(rs.choice([32, 64, 128, 160, 512, 768]))
        K = int(rs.choice([64, 128, 192, 256, 512, 768]))
        per_channel = bool(rs.randint(2))
        bw, ba = int(rs.choice([4, 8])), int(rs.choice([4, 6, 8]))
        g = torch.Generator().manual_seed(int(rs.randint(1 << 30)))
        w = torch.randn(N, K, generator=g) * 0.05
        x = torch.randn(M, K, generator=g) * 1.5 + 0.3
        b = torch.randn(N, generator=g) * 0.1
        wd, _ = O.sym_params_from_range(w.amin(1) if per_channel else w.min(), w.amax(1) if per_channel else w.max(), bw)
        xd, xz = O.asym_params_from_range(x.min(), x.max(), ba)
        x_i8 = be.quantize_to_int8(x.cuda(), xd.cuda(), xz.cuda(), None, ba, False, False, 1e-8, 1, 1, minus_128=True)
        w_i8 = be.quantize_to_int8(w.cuda(), wd.cuda(), None, torch.tensor(True).cuda(), bw, True, False, 1e-8,
                                   N if per_channel else 1, K if per_channel else 1, minus_128=False)
        rsum = be.rowsum_i8(w_i8)
        xq = (xd.cuda(), xz.cuda(), ba, 1e-8)
        wdd = wd.cuda().reshape(-1)
        lin = be.linear_i8(x_i8, w_i8, rsum, b.cuda(), xq, wdd, 1e-8, _hip.ACT_NONE, None, torch.float32)
        s = float(lin.abs().max()) + 1e-3

        def q7(lo, hi, bits):
            d, z = O.asym_params_from_range(lo, hi, bits)
            return (d.cuda(), z.cuda(), None, bits, False, False, 1e-8)
        tag = (seed, case, M, N, K, per_channel, bw, ba)
        # -- output quantizer (+ indices), no activation / ReLU: bit-exact
        qo = q7(-rs.uniform(0.3, 1.1) * s, rs.uniform(0.3, 1.1) * s, int(rs.choice([4, 8])))
        for act, fn in ((_hip.ACT_NONE, lambda v: v), (_hip.ACT_RELU, torch.relu)):
            y, yi = be.linear_i8(x_i8, w_i8, rsum, b.cuda(), xq, wdd, 1e-8, act, qo, torch.float32, want_idx=True)
            ref_y, ref_i = be.fake_quant(fn(lin), *qo, 1, 1, idx_dtype=torch.float32)
            assert torch.equal(y, ref_y), tag
            assert torch.equal(yi.int() + 128, ref_i.int()), tag
        # -- GELU
        yg = be.linear_i8(x_i8, w_i8, rsum, b.cuda(), xq, wdd, 1e-8, _hip.ACT_GELU, None, torch.float32)
        ref_g = torch.nn.functional.gelu(lin)
        assert float((yg - ref_g).abs().max()) <= 2.4e-7 * max(s, 1.0), tag
        ygq = be.linear_i8(x_i8, w_i8, rsum, b.cuda(), xq, wdd, 1e-8, _hip.ACT_GELU, qo, torch.float32)
        same = (ygq == be.fake_quant(ref_g, *qo, 1, 1)[0]).float().mean().item()
        assert same >= 0.9995, (tag, same)
        # -- NoNorm tail
        with_res = bool(rs.randint(2))
        res = (torch.randn(M, N, generator=g) * 0.5 * s).cuda() if with_res else None
        nw = (1 + 0.3 * torch.randn(N, generator=g)).cuda()
        nb = (0.2 * s * torch.randn(N, generator=g)).cuda()
        q_dense = q7(-0.8 * s, 0.9 * s, int(rs.choice([4, 8]))) if rs.randint(3) else None
        q_sum = q7(-1.5 * s, 1.4 * s, 8) if (with_res and rs.randint(3)) else None
        q_out = q7(-1.6 * s, 1.7 * s, int(rs.choice([4, 8]))) if rs.randint(4) else None
        t = lin if q_dense is None else be.fake_quant(lin, *q_dense, 1, 1)[0]
        if with_res:
            t = t + res
            if q_sum is not None:
                t = be.fake_quant(t, *q_sum, 1, 1)[0]
        t = t * nw + nb
        if q_out is not None:
            ref_y, ref_i = be.fake_quant(t, *q_out, 1, 1, idx_dtype=torch.float32)
        else:
            ref_y, ref_i = t, None
        got = be.linear_i8_nonorm(x_i8, w_i8, rsum, b.cuda(), res, nw, nb, xq, wdd, 1e-8, q_dense, q_sum, q_out,
                                  torch.float32, want_idx=q_out is not None)
        got_y, got_i = got if isinstance(got, tuple) else (got, None)
        assert torch.equal(got_y, ref_y), (tag, 'tail', with_res, q_dense is None, q_sum is None, q_out is None)
        if ref_i is not None:
            assert torch.equal(got_i.int() + 128, ref_i.int()), (tag, 'tail idx')
