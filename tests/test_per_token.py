"""Per-token (`--per-token`: axis = 1 on [B, T, d] / [B, T, D]) and dynamic (`--dynamic`) activation quantization at
BERT-base shapes against tests/golden/per_token.npz -- outputs of the REFERENCE's QuantizationManager
(tests/golden/make_golden_per_token.py; reference main.py:359-376, utils/per_embd_quant_utils.py:54-68,
quantization_manager.py:99-106).  Dynamic mode is the one mode where estimate + quantize run on every inference call.

* CPU: the oracle backend through the drop-in manager reproduces all 18 cases bit for bit (ranges, parameters, SHA-256
  of the index tensor and of the output, first / last token rows in full).
* GPU: the same through the HIP kernels (`mm_rows_wave`, `fq_rows_wave`, the fused `tq_calibrate_minmax`), both
  calibration routes; plus ragged / multi-slice launch shapes against the oracle.
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from tests.conftest import GOLDEN


def make_input(seed, shape, outlier_dims=(308, 381), scale=20.0):
    """== tests/golden/make_golden_per_token.py::make_input (numpy MT19937 stream: build independent)."""
    rs = np.random.RandomState(seed)
    x = rs.standard_normal(size=shape).astype(np.float32)
    for d in outlier_dims:
        if d < shape[-1]:
            x[..., d] *= scale
    x *= (1.0 + np.arange(shape[1], dtype=np.float32) / shape[1])[None, :, None]
    return x


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _fixture():
    z = np.load(os.path.join(GOLDEN, 'per_token.npz'))
    return z, json.loads(str(z['meta']))


def _run_case(m, z, device):
    from quantization.quantization_manager import QuantizationManager
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from utils.per_embd_quant_utils import set_act_quant_axis_and_groups
    k, shape = m['k'], tuple(m['shape'])
    dt = torch.bfloat16 if m['io'] == 'bf16' else torch.float32
    x = torch.from_numpy(make_input(m['seed'], shape)).to(dt).to(device)
    mgr = QuantizationManager(qmethod=QMethods.asymmetric_uniform, init=RangeEstimators.current_minmax,
                              qparams=dict(n_bits=m['n_bits']))
    set_act_quant_axis_and_groups(mgr, axis=1, n_groups=None)
    if m['mode'] == 'fixed':
        y0 = mgr(x)
        mgr.fix_ranges()
        y = mgr(x)
        assert torch.equal(y, y0), m
    else:
        mgr(torch.from_numpy(make_input(m['seed'] + 500, shape)).to(dt).to(device))
        y = mgr(x)                                   # dynamic: still estimating, ranges follow THIS input
    q = mgr.quantizer
    est = mgr.range_estimator
    assert y.dtype == dt
    assert np.array_equal(est.current_xmin.cpu().numpy().reshape(-1), z[f'p{k}_xmin']), m
    assert np.array_equal(est.current_xmax.cpu().numpy().reshape(-1), z[f'p{k}_xmax']), m
    assert np.array_equal(q._delta.cpu().numpy().reshape(-1), z[f'p{k}_delta']), m
    assert np.array_equal(q._zero_float.cpu().numpy().reshape(-1), z[f'p{k}_zero_float']), m
    idx = q.to_integer_forward(x).cpu().numpy()
    idx_u8 = idx.astype(np.uint8)
    assert np.array_equal(idx_u8.astype(np.float32), idx), m
    y_out = y.cpu().view(torch.int16).numpy() if m['io'] == 'bf16' else y.cpu().numpy()
    T = shape[1]
    assert np.array_equal(idx_u8[:, [0, T - 1], :], z[f'p{k}_idx_rows']), m
    assert np.array_equal(y_out[:, [0, T - 1], :], z[f'p{k}_y_rows']), m
    assert sha(idx_u8) == m['idx_sha256'], m
    assert sha(y_out) == m['y_sha256'], m


def test_per_token_and_dynamic_cpu_oracle_equals_reference():
    from quantization import _hip
    from tests._oracle_backend import OracleBackend
    z, meta = _fixture()
    assert len(meta) == 18
    prev = _hip.set_backend(OracleBackend())
    try:
        for m in meta:
            _run_case(m, z, 'cpu')
    finally:
        _hip.set_backend(prev)


@pytest.mark.gpu
@pytest.mark.parametrize('fused', [True, False], ids=['fused-calibration', 'layered-calibration'])
def test_per_token_and_dynamic_gpu_equals_reference(fused):
    from quantization import _hip, quantization_manager as qm
    assert _hip.backend().name == 'hip'
    z, meta = _fixture()
    prev = qm.FUSED_CALIBRATION
    qm.FUSED_CALIBRATION = fused
    try:
        for m in meta:
            _run_case(m, z, 'cuda')
    finally:
        qm.FUSED_CALIBRATION = prev


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape,axis', [
    ((8, 128, 768), 1), ((8, 128, 3072), 1),        # BERT-base [B, T, d] and [B, T, D] per token
    ((9, 4096, 16), 1),                             # 2 slices of the outer index: the 4-row batches AND their remainder
    ((70, 128, 64), 1),                             # 64 slices, some with two rows
    ((5, 40, 24), 1), ((3, 7, 8), 1),               # ragged: rows shorter than a wave
    ((3072, 768), 0),                               # per-channel weight layout (outer = 1)
    ((4, 6, 4104), 1),                              # rows longer than the wave kernels take: the block-per-row kernels
    ((131, 500, 136), 1),                           # 17 / 34 vectors per row: the flat U-row batches with ragged last iteration
    ((19, 33, 1000), 1),                            # 125 / 250 vectors per row: the statistics walk 4 rows as one index space
])
@pytest.mark.parametrize('rows_flat', [None, '2', '3'], ids=['by-size', 'flat-per-lane', 'flat-table'])
def test_row_parameter_launch_shapes_vs_oracle(dtype, shape, axis, rows_flat, monkeypatch):
    """Statistics, parameters, indices and outputs for row-parameter layouts through the wave-per-(parameter, slice)
    kernels and their neighbours, against the oracle on the same tensor.  TQ_ROWS_FLAT = 2 / 3: the same through the
    flat-tile kernels that large launches take (fq_rows_flat: parameters per lane; fq_rows_tab: per tile into LDS, rows of
    >= 4 vectors)."""
    if rows_flat is not None:
        monkeypatch.setenv('TQ_ROWS_FLAT', rows_flat)
    from oracle import tq_oracle as O
    from quantization import _hip
    be = _hip.backend()
    g = torch.Generator().manual_seed(sum(shape) + axis)
    x = (torch.randn(*shape, generator=g) * torch.linspace(0.5, 4, shape[axis]).view(
        [-1 if i == axis else 1 for i in range(len(shape))])).to(dtype)
    xf = x.float()
    inner = int(np.prod(shape[axis + 1:]))
    xd = x.cuda()
    mn, mx = be.minmax(xd, shape[axis], inner)
    rmn, rmx = O.minmax_axis(xf, axis)
    assert torch.equal(mn.cpu(), rmn) and torch.equal(mx.cpu(), rmx)
    for n_bits in (8, 4):
        delta, zf = O.asym_params_from_range(rmn, rmx, n_bits)
        d2, z2 = be.set_range_asym(mn, mx, n_bits, 1e-8, False)
        assert torch.equal(d2.cpu(), delta) and torch.equal(z2.cpu(), zf)
        ref_idx, ref_y = O.fake_quant_lowp(x, delta, zf, n_bits, False, axis=axis)
        y, idx = be.fake_quant(xd, d2, z2, None, n_bits, False, False, 1e-8, shape[axis], inner, idx_dtype=torch.float32)
        assert torch.equal(idx.cpu(), ref_idx)
        assert torch.equal(y.cpu(), ref_y)
        y8, idx8 = be.fake_quant(xd, d2, z2, None, n_bits, False, False, 1e-8, shape[axis], inner, idx_dtype=torch.uint8)
        assert torch.equal(idx8.cpu().float(), ref_idx) and torch.equal(y8.cpu(), ref_y)
    # NaN in one row poisons that row's range only (torch.min / torch.max semantics)
    xn = xd.clone()
    xn.view(-1)[inner * 2 + 1] = float('nan')
    mn2, mx2 = be.minmax(xn, shape[axis], inner)
    rmn2, rmx2 = O.minmax_axis(xn.float().cpu(), axis)
    assert torch.equal(torch.isnan(mn2.cpu()), torch.isnan(rmn2)) and torch.equal(torch.isnan(mx2.cpu()), torch.isnan(rmx2))
    ok = ~torch.isnan(rmn2)
    assert torch.equal(mn2.cpu()[ok], rmn2[ok]) and torch.equal(mx2.cpu()[ok], rmx2[ok])


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('init', ['current_minmax', 'running_minmax', 'allminmax'])
@pytest.mark.parametrize('shape', [(8, 128, 768), (3, 40, 24), (2, 5, 8), (21, 16, 768), (8, 16, 3072)])
def test_one_pass_dynamic_step_equals_the_separate_launches(dtype, init, shape):
    """`tq_calibrate_minmax` runs row-parameter layouts whose per-parameter data fits in a block's registers as ONE launch
    (calib_rows_onepass_k: statistics, estimator rule, parameters, quantization, one read of x).  Three batches through the
    manager on the fused route against the layered route (tq_minmax -> tq_range_update -> tq_set_range_asym ->
    tq_fake_quant_fwd): estimator state, parameters and outputs equal bit for bit after every batch -- incl. a NaN in one
    token's rows (that token's range turns NaN, the others do not), shapes at both sides of the one-pass limit, and the
    in-place state of options.INPLACE_CALIBRATION_STATE."""
    from quantization import options, quantization_manager as qm
    from quantization.quantization_manager import QuantizationManager
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from utils.per_embd_quant_utils import set_act_quant_axis_and_groups
    if init == 'allminmax':
        pytest.skip('AllMinMax ignores the axis upstream (quirk q5): per-tensor statistics, not this kernel')
    xs = [torch.from_numpy(make_input(900 + i, shape, outlier_dims=(3, 5))).to(dtype).cuda() * (1 + 0.5 * i) for i in range(3)]
    xs[2].view(-1)[shape[-1] * 2 + 1] = float('nan')               # row of token 2 (batch entry 0)

    def run(fused, inplace):
        prev, prev_inp = qm.FUSED_CALIBRATION, options.INPLACE_CALIBRATION_STATE
        qm.FUSED_CALIBRATION, options.INPLACE_CALIBRATION_STATE = fused, inplace
        try:
            mgr = QuantizationManager(qmethod=QMethods.asymmetric_uniform, init=RangeEstimators[init], qparams=dict(n_bits=8))
            set_act_quant_axis_and_groups(mgr, axis=1, n_groups=None)
            out = []
            for x in xs:
                y = mgr(x)
                out.append((y.clone(), mgr.range_estimator.current_xmin.clone().reshape(-1), mgr.range_estimator.current_xmax.clone().reshape(-1),
                            mgr.quantizer._delta.clone().reshape(-1), mgr.quantizer._zero_float.clone().reshape(-1)))
            return out
        finally:
            qm.FUSED_CALIBRATION, options.INPLACE_CALIBRATION_STATE = prev, prev_inp

    ref = run(False, False)
    for inplace in (False, True):
        got = run(True, inplace)
        for b, (r, g) in enumerate(zip(ref, got)):
            for k, (rt, gt) in enumerate(zip(r, g)):
                same = torch.equal(rt.view(torch.int16 if rt.element_size() == 2 else torch.int32),
                                   gt.view(torch.int16 if gt.element_size() == 2 else torch.int32))
                nan_same = torch.equal(torch.isnan(rt), torch.isnan(gt)) and torch.equal(rt[~torch.isnan(rt)], gt[~torch.isnan(gt)])
                assert same or nan_same, (init, shape, dtype, inplace, b, k)
    assert torch.isnan(ref[2][1]).sum() == 1                        # exactly one token position poisoned


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_misaligned_and_odd_row_lengths_take_the_fallback_kernels(dtype):
    """The wave / one-pass kernels need 16-byte aligned tensors and rows that are whole 16-byte vectors; anything else must
    fall back (scalar / block-per-row kernels, separate launches) and give the SAME statistics, parameters and outputs as
    an aligned copy of the same values."""
    from quantization import _hip
    be = _hip.backend()
    for shape in ((8, 16, 24), (4, 10, 7), (3, 5, 33)):
        n = int(np.prod(shape))
        g = torch.Generator().manual_seed(n)
        vals = (torch.randn(n, generator=g) * 3).to(dtype)
        aligned = vals.clone().cuda().view(shape)
        holder = torch.empty(n + 1, dtype=dtype, device='cuda')
        holder[1:] = vals.cuda()
        skewed = holder[1:].view(shape)                       # same values, data_ptr % 16 != 0
        assert skewed.data_ptr() % 16 != 0 and skewed.is_contiguous()
        T, inner = shape[1], shape[2]
        ref = be.calibrate_minmax(aligned, T, inner, _hip.EST_CURRENT, None, None, 0.9, 0, None, 8, False, 1e-8, False)
        got = be.calibrate_minmax(skewed, T, inner, _hip.EST_CURRENT, None, None, 0.9, 0, None, 8, False, 1e-8, False)
        for r, t in zip(ref, got):
            if r is not None:
                assert torch.equal(r, t), shape
        mn_a, mx_a = be.minmax(aligned, T, inner)
        mn_s, mx_s = be.minmax(skewed, T, inner)
        assert torch.equal(mn_a, mn_s) and torch.equal(mx_a, mx_s)
        ya, ia = be.fake_quant(aligned, ref[2], ref[3], None, 8, False, False, 1e-8, T, inner, idx_dtype=torch.uint8)
        ys, is_ = be.fake_quant(skewed, ref[2], ref[3], None, 8, False, False, 1e-8, T, inner, idx_dtype=torch.uint8)
        assert torch.equal(ya, ys) and torch.equal(ia, is_) and torch.equal(ya, ref[5])


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_per_token_full_size_properties(dtype):
    """BASELINE's full activation size with per-token ranges ([256,512,768]: 201 / 403 MB, above the size from which the
    flat-tile kernel fq_rows_tab takes row-parameter launches): statistics == torch's reductions, idempotence on the fixed
    grid, indices on the grid and consistent with the output, the wave-per-row route (TQ_ROWS_FLAT=0) bit-identical, and
    slabs at both ends and in the middle of the tensor against the oracle."""
    from oracle import tq_oracle as O
    from quantization import _hip
    be = _hip.backend()
    B, T, D = 256, 512, 768
    torch.manual_seed(77)
    x = torch.randn(B, T, D, device='cuda', dtype=dtype)
    x[..., 308] *= 20
    x *= torch.linspace(0.5, 3.0, T, device='cuda', dtype=dtype).view(1, T, 1)
    mn, mx = be.minmax(x, T, D)
    xt = x.float().transpose(0, 1).reshape(T, -1)
    assert torch.equal(mn, xt.amin(1)) and torch.equal(mx, xt.amax(1))
    d, z = be.set_range_asym(mn, mx, 8, 1e-8, False)
    y, idx = be.fake_quant(x, d, z, None, 8, False, False, 1e-8, T, D, idx_dtype=torch.uint8)
    os.environ['TQ_ROWS_FLAT'] = '0'
    try:
        y0, idx0 = be.fake_quant(x, d, z, None, 8, False, False, 1e-8, T, D, idx_dtype=torch.uint8)
    finally:
        os.environ.pop('TQ_ROWS_FLAT')
    assert torch.equal(y, y0) and torch.equal(idx, idx0)
    y2, _ = be.fake_quant(y, d, z, None, 8, False, False, 1e-8, T, D)
    assert torch.equal(y2, y)                                              # Q(Q(x)) == Q(x)
    _, idx_only = be.fake_quant(x, d, z, None, 8, False, False, 1e-8, T, D, want_y=False, idx_dtype=torch.uint8)
    assert torch.equal(idx_only, idx)
    zp = torch.round(z).clamp(0, 255).view(1, T, 1)
    deq = (d.view(1, T, 1) * (idx.float() - zp)).to(dtype)
    assert torch.equal(deq, y)
    for b in (0, B // 2 + 1, B - 1):
        ref_idx, ref_y = O.fake_quant_lowp(x[b:b + 1].cpu(), d.cpu(), z.cpu(), 8, False, axis=1)
        assert torch.equal(idx[b:b + 1].cpu().float(), ref_idx) and torch.equal(y[b:b + 1].cpu(), ref_y)
