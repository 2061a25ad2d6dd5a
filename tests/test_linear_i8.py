"""(f3) fused integer Linear + bias + activation + fake-quant on the i8 matrix cores.

The integer contraction is exact; the reference's fp32 simulation `F.linear(Q(x), Q(W), b)` is an
fp32-rounded evaluation of the same number.  Bars: pre-quantizer output within 1e-5 (relative to the
row scale) of the CPU fp32 simulation and within fp32 epsilon of the float64 value; after the output
quantizer >= 99.9 % of elements identical to the oracle chain, the rest one grid step away."""
import os

import numpy as np
import pytest
import torch

from oracle import tq_oracle as O

pytestmark = pytest.mark.gpu


def _problem(M, N, K, n_bits_w, n_bits_a, per_channel, seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(N, K, generator=g) * 0.05
    x = torch.randn(M, K, generator=g) * 1.5 + 0.3
    b = torch.randn(N, generator=g) * 0.1
    if per_channel:
        wd, ws = O.sym_params_from_range(w.amin(1), w.amax(1), n_bits_w)
    else:
        wd, ws = O.sym_params_from_range(w.min(), w.max(), n_bits_w)
    assert bool(ws)
    xd, xz = O.asym_params_from_range(x.min(), x.max(), n_bits_a)
    w_idx, w_q = O.fake_quant(w, wd, None, n_bits_w, True, True, per_channel=per_channel)
    x_idx, x_q = O.fake_quant(x, xd, xz, n_bits_a, False)
    return dict(w=w, x=x, b=b, wd=wd, xd=xd, xz=xz, w_idx=w_idx, w_q=w_q, x_idx=x_idx, x_q=x_q)


@pytest.mark.parametrize('shape', [(1024, 768, 768), (1024, 3072, 768), (1024, 768, 3072), (64, 128, 512),
                                   (1024, 512, 384), (32, 32, 64),
                                   (4096, 4096, 128),      # 128 x 128 block tiles (>= 1024 of them)
                                   (96, 160, 192),         # M, N % 64 != 0: LDS-free kernel
                                   (64, 64, 16384),        # longest supported K (i32 accumulator bound)
                                   (128, 64, 64)])         # K % 128 != 0: LDS-free kernel
@pytest.mark.parametrize('cfg', [(8, 8, False), (4, 4, False), (8, 8, True), (4, 8, True)])
def test_linear_i8_vs_fp32_simulation(shape, cfg):
    from quantization import _hip
    be = _hip.backend()
    M, N, K = shape
    n_bits_w, n_bits_a, per_channel = cfg
    p = _problem(M, N, K, n_bits_w, n_bits_a, per_channel, seed=M + N + K)
    dev = lambda t: t.cuda()
    x_i8 = be.quantize_to_int8(dev(p['x_q']), dev(p['xd']), dev(p['xz']), None, n_bits_a, False, False, 1e-8,
                               1, 1, minus_128=True)
    assert torch.equal(x_i8.cpu().int() + 128, p['x_idx'].int())          # re-quantisation is exact
    n_par = N if per_channel else 1
    w_i8 = be.quantize_to_int8(dev(p['w_q']), dev(p['wd']), None, dev(torch.tensor(True)), n_bits_w, True, False,
                               1e-8, n_par, K if per_channel else 1, minus_128=False)
    assert torch.equal(w_i8.cpu().int(), p['w_idx'].int())
    rs = be.rowsum_i8(w_i8)
    assert torch.equal(rs.cpu().long(), p['w_idx'].long().sum(1))
    sim32 = torch.nn.functional.linear(p['x_q'], p['w_q'], p['b'])                      # the reference's path
    sim64 = torch.nn.functional.linear(p['x_q'].double(), p['w_q'].double(), p['b'].double())
    xq = (dev(p['xd']), dev(p['xz']), n_bits_a, 1e-8)
    y = be.linear_i8(x_i8, w_i8, rs, dev(p['b']), xq, dev(p['wd']).reshape(-1), 1e-8, _hip.ACT_NONE, None,
                     torch.float32).cpu()
    scale = sim64.abs().max().item()
    assert (y.double() - sim64).abs().max().item() <= 4e-7 * scale + 1e-7     # fp32 epsilon of the exact value
    assert (y - sim32).abs().max().item() <= 1e-5 * scale                      # north-star tolerance
    # branch-free GELU of the fast epilogue vs the float64 value, and vs the generic (libm erff) epilogue
    yg = be.linear_i8(x_i8, w_i8, rs, dev(p['b']), xq, dev(p['wd']).reshape(-1), 1e-8, _hip.ACT_GELU, None,
                      torch.float32).cpu()
    assert (yg.double() - torch.nn.functional.gelu(sim64)).abs().max().item() <= 6e-7 * scale + 2e-7
    os.environ['TQ_I8_FAST_EPI'] = '0'
    try:
        yg0 = be.linear_i8(x_i8, w_i8, rs, dev(p['b']), xq, dev(p['wd']).reshape(-1), 1e-8, _hip.ACT_GELU, None,
                           torch.float32).cpu()
    finally:
        del os.environ['TQ_I8_FAST_EPI']
    assert (yg - yg0).abs().max().item() <= 2.4e-7 * max(scale, 1.0)
    # with activation + output quantizer
    for act, fn in ((_hip.ACT_GELU, torch.nn.GELU()), (_hip.ACT_RELU, torch.relu), (_hip.ACT_TANH, torch.tanh),
                    (_hip.ACT_NONE, lambda v: v)):
        pre = fn(sim64).float()
        od, oz = O.asym_params_from_range(pre.min(), pre.max(), 8)
        _, ref = O.fake_quant(pre, od, oz, 8, False)
        q_out = (dev(od), dev(oz), None, 8, False, False, 1e-8)
        yq = be.linear_i8(x_i8, w_i8, rs, dev(p['b']), xq, dev(p['wd']).reshape(-1), 1e-8, act, q_out,
                          torch.float32).cpu()
        diff = (yq - ref).abs()
        assert (diff == 0).float().mean().item() >= 0.999, (act, (diff == 0).float().mean().item())
        assert diff.max().item() <= float(od) * 1.001
        # the optional int8 index output is consistent with the dequantised output it accompanies
        y2, yi = be.linear_i8(x_i8, w_i8, rs, dev(p['b']), xq, dev(p['wd']).reshape(-1), 1e-8, act, q_out,
                              torch.float32, want_idx=True)
        assert torch.equal(y2.cpu(), yq)
        zp = O.effective_zero_point(oz, 8)
        assert torch.equal((yi.cpu().float() + 128 - zp) * od, yq)
    yb = be.linear_i8(x_i8, w_i8, rs, dev(p['b']), xq, dev(p['wd']).reshape(-1), 1e-8, _hip.ACT_NONE, None,
                      torch.bfloat16).cpu()
    assert torch.equal(yb, y.to(torch.bfloat16))


@pytest.mark.parametrize('shape', [(8192, 768, 768), (8192, 768, 3072), (4096, 3072, 768), (2048, 768, 1024)])
def test_block_tile_sizes_are_bit_identical(shape):
    """128 x 128 block tiles are taken from 384 tiles on for K >= 512 (round 6; 1024 tiles before): the shapes a BERT-base
    layer runs at 4096 / 8192 tokens on both sides of that rule, forced onto either tile size (TQ_I8_BIG_MIN = 1 / huge) --
    pre-quantizer fp32 output, quantized output and int8 indices equal bit for bit (the contraction is exact integer
    arithmetic and the epilogue is the same per output), and the default choice equals them."""
    from quantization import _hip
    be = _hip.backend()
    M, N, K = shape
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    x_i8 = torch.randint(-128, 128, (M, K), dtype=torch.int8, device='cuda', generator=g)
    w_i8 = torch.randint(-127, 128, (N, K), dtype=torch.int8, device='cuda', generator=g)
    rs = be.rowsum_i8(w_i8)
    b = torch.randn(N, device='cuda', generator=g)
    xq = (torch.tensor(0.02, device='cuda'), torch.tensor(117.0, device='cuda'), 8, 1e-8)
    wd = torch.tensor([0.001], device='cuda')
    q_out = (torch.tensor(0.05, device='cuda'), torch.tensor(100.0, device='cuda'), None, 8, False, False, 1e-8)

    def run():
        y0 = be.linear_i8(x_i8, w_i8, rs, b, xq, wd, 1e-8, _hip.ACT_NONE, None, torch.float32)
        y1, i1 = be.linear_i8(x_i8, w_i8, rs, b, xq, wd, 1e-8, _hip.ACT_GELU, q_out, torch.float32, want_idx=True)
        return y0, y1, i1
    base = run()
    outs = []
    for big_min in ('1', '100000000'):
        os.environ['TQ_I8_BIG_MIN'] = big_min
        try:
            outs.append(run())
        finally:
            del os.environ['TQ_I8_BIG_MIN']
    for a, c, d in zip(base, *outs):
        assert torch.equal(a, c) and torch.equal(a, d)
    # a slab against the exact integer contraction (fp32 epsilon of the exact value)
    acc = x_i8[:64].double() @ w_i8.double().t()
    exact = (acc - (117.0 - 128.0) * w_i8.double().sum(1)) * (0.02 * 0.001) + b.double()
    assert (base[0][:64].double() - exact).abs().max().item() <= 4e-7 * exact.abs().max().item() + 1e-7


@pytest.mark.parametrize('shape', [(1024, 768, 768), (1024, 768, 3072), (64, 64, 256), (128, 192, 512), (256, 256, 1024),
                                   (64, 128, 1280), (128, 64, 2048), (1024, 1024, 1792), (64, 64, 16384)])
@pytest.mark.parametrize('act', ['none', 'gelu'])
def test_ring_kernel_equals_the_double_buffered_kernel(shape, act):
    """Launches of at most one 64 x 64 tile per CU with K >= 1024 run the 8-stage slab ring (csrc/tq_linear_i8.hip,
    NS = 8), those with a shorter K 32 x 32 tiles: same integer contraction, same epilogue -- both bit-identical to the
    double-buffered 64 x 64 kernel (TQ_I8_RING_MAX_GRID=0, TQ_I8_SMALL_MAX_GRID=0), for K of fewer / as many / more slabs
    than the ring holds (TQ_I8_RING_MIN_K=256 sends the short K through the ring as well)."""
    from quantization import _hip
    be = _hip.backend()
    M, N, K = shape
    p = _problem(M, N, K, 8, 8, True, seed=K + N)
    dev = lambda t: t.cuda()
    x_i8 = be.quantize_to_int8(dev(p['x_q']), dev(p['xd']), dev(p['xz']), None, 8, False, False, 1e-8, 1, 1, minus_128=True)
    w_i8 = be.quantize_to_int8(dev(p['w_q']), dev(p['wd']), None, dev(torch.tensor(True)), 8, True, False, 1e-8, N, K,
                               minus_128=False)
    rs = be.rowsum_i8(w_i8)
    xq = (dev(p['xd']), dev(p['xz']), 8, 1e-8)
    a = _hip.ACT_GELU if act == 'gelu' else _hip.ACT_NONE
    pre = torch.nn.functional.linear(p['x_q'], p['w_q'], p['b'])
    pre = torch.nn.functional.gelu(pre) if act == 'gelu' else pre
    od, oz = O.asym_params_from_range(pre.min(), pre.max(), 8)
    q_out = (dev(od), dev(oz), None, 8, False, False, 1e-8)

    def run():
        y0 = be.linear_i8(x_i8, w_i8, rs, dev(p['b']), xq, dev(p['wd']).reshape(-1), 1e-8, a, None, torch.float32)
        y1, i1 = be.linear_i8(x_i8, w_i8, rs, dev(p['b']), xq, dev(p['wd']).reshape(-1), 1e-8, a, q_out, torch.float32,
                              want_idx=True)
        return y0.cpu(), y1.cpu(), i1.cpu()
    os.environ['TQ_I8_RING_MIN_K'] = '256'
    try:
        ring = run()
    finally:
        del os.environ['TQ_I8_RING_MIN_K']
    os.environ['TQ_I8_RING_MAX_GRID'] = '0'              # -> 32 x 32 tiles (at most one 64 x 64 tile per CU), double buffer
    try:
        small = run()
        os.environ['TQ_I8_SMALL_MAX_GRID'] = '0'         # -> 64 x 64 tiles, double buffer
        try:
            plain = run()
        finally:
            del os.environ['TQ_I8_SMALL_MAX_GRID']
    finally:
        del os.environ['TQ_I8_RING_MAX_GRID']
    for r, q, t in zip(ring, plain, small):
        assert torch.equal(r, q) and torch.equal(t, q)
    # and against the exact integer contraction
    acc = p['x_idx'].double() @ p['w_idx'].double().t()
    zx = O.effective_zero_point(p['xz'], 8)
    exact = (acc - zx * p['w_idx'].double().sum(1)) * (p['xd'].double() * p['wd'].double().reshape(1, -1)) + p['b'].double()
    if act == 'none':
        assert (ring[0].double() - exact).abs().max().item() <= 4e-7 * exact.abs().max().item() + 1e-7


@pytest.mark.layered_route
def test_bert_forward_with_integer_linears():
    """Whole BERT-base fixed-range forward with every eligible Linear on the i8 matrix cores (and the
    fused layer tails): logits stay within the same envelope as CPU-vs-GPU GEMM round-off."""
    from quantization import options
    from tests.test_bert_e2e import _build, _fixture, _calibrate_and_run
    from tests.harness_bert import QResidualBlock
    z = _fixture()
    model, _ = _build('cuda')
    ids = torch.from_numpy(z['input_ids']).cuda()
    layered = _calibrate_and_run(model, ids)
    calls = {'n': 0}
    from quantization import _hip
    be = _hip.backend()
    orig = be.linear_i8

    def counting(*a, **k):
        calls['n'] += 1
        return orig(*a, **k)
    be.linear_i8 = counting
    options.INT8_LINEAR = True
    try:
        with torch.no_grad():
            y_int = model(ids)
            n_plain = calls['n']
            QResidualBlock.fuse = True
            y_int_fused = model(ids)
    finally:
        options.INT8_LINEAR = False
        QResidualBlock.fuse = False
        be.__dict__.pop('linear_i8', None)        # drop the instance attribute again (other tests patch the class)
    assert n_plain == 12 * 6                      # q, k, v, attention-out, intermediate, output per layer
    # the 12 intermediate Linears (GELU) went through staircase tables, one per layer, built on the device; the verdict
    # of each builder is in its table (a grid too fine for the bins keeps the arithmetic epilogue: also a valid outcome)
    from quantization.autoquant_utils import int8_stair_status
    tables = int8_stair_status(model)
    assert len(tables) == 12 and all('intermediate' in n for n in tables), list(tables)
    print('GELU staircase tables accepted by their builders: %d of 12' % sum(all(v.values()) for v in tables.values()))
    span = float(layered.max() - layered.min())
    for y in (y_int, y_int_fused):
        assert float((y - layered).abs().max()) <= 0.08 * span           # measured 5-7 % (round 3), bar tightened from 10 %
    # per encoder layer, on the SAME input: the fraction of 8-bit output indices the integer evaluation moves
    from harness.divergence import encoder_flip_rates

    class _Int8:
        def __enter__(self):
            options.INT8_LINEAR = True

        def __exit__(self, *exc):
            options.INT8_LINEAR = False
            return False
    rows, first = encoder_flip_rates(model, ids, _Int8())
    same = [r['same_input']['flip_rate'] for r in rows]
    print('BERT-base W8A8 integer Linears, same-input flip rate per layer: max %.2e, max index distance %d; first diverging '
          'layer (free-running): %s' % (max(same), max(r['same_input']['max_steps'] for r in rows), first))
    # (measured: max 6e-4 of a layer's 786 432 outputs, index distance <= 2)
    assert max(same) <= 3e-3 and max(r['same_input']['max_steps'] for r in rows) <= 2, same


def test_integer_linear_in_training_mode_qat_forward():
    """BASELINE config 4 ("QAT forward, fused Linear+quant MFMA path"): with fixed ranges a TRAINING-mode forward under
    autograd runs on the integer matrix-core kernel (options.INT8_LINEAR) and its backward is the straight-through
    estimator of the layered modules (reference hijacker.py:66-116, quantizers.py:12-33).  Bars: outputs >= 99.9 %
    identical to the layered fp32-simulation forward (rest one grid step of the output quantizer); for the SAME upstream
    gradient the gradients w.r.t. input, weight and bias are bit-identical to the layered path's."""
    from quantization import options
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from quantization.base_quantized_model import QuantizedModel
    from quantization.base_quantized_classes import QuantizedActivation
    from quantization.autoquant_utils import quantize_model

    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=4, n_bits_act=4,
              weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)

    class Net(QuantizedModel):
        def __init__(self):
            super().__init__()
            torch.manual_seed(5)
            self.inp = QuantizedActivation(**qp)
            self.fc1 = quantize_model(torch.nn.Sequential(torch.nn.Linear(512, 512), torch.nn.ReLU()), **qp)
            self.fc2 = quantize_model(torch.nn.Linear(512, 128), **qp)

        def forward(self, x):
            return self.fc2(self.fc1(self.inp(x)))

    net = Net().cuda()
    g = torch.Generator().manual_seed(6)
    x = (torch.randn(8, 128, 512, generator=g) * 1.3).cuda()
    net.set_quant_state(True, True)
    net.eval()
    with torch.no_grad():
        net(x)
    net.fix_ranges()
    net.train()
    gy = torch.randn(8, 128, 128, generator=g).cuda()

    def run(int8):
        options.INT8_LINEAR = int8
        try:
            for p in net.parameters():
                p.grad = None
            xr = x.clone().requires_grad_(True)
            y = net(xr)
            y.backward(gy)
            return y.detach(), xr.grad.clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        finally:
            options.INT8_LINEAR = False

    from quantization.autoquant_utils import INT8_STATS
    before = dict(INT8_STATS)
    y_i, gx_i, gp_i = run(True)
    assert INT8_STATS['autograd_calls'] - before['autograd_calls'] == 2 and \
        INT8_STATS['kernel_calls'] - before['kernel_calls'] == 2, \
        'both Linears must have run on the integer MFMA kernel in training mode'
    y_l, gx_l, gp_l = run(False)
    step = float(net.fc2.activation_quantizer.quantizer._delta)
    diff = (y_i - y_l).abs()
    assert float((diff == 0).float().mean()) >= 0.999 and float(diff.max()) <= step * 1.001
    # the second layer's backward sees the same grad_y and (by construction) the layered graph of that layer
    assert torch.equal(gp_i['fc2.weight'], gp_l['fc2.weight']) or torch.allclose(gp_i['fc2.weight'], gp_l['fc2.weight'], rtol=1e-3, atol=1e-5)
    assert set(gp_i) == set(gp_l) and all(torch.isfinite(v).all() for v in gp_i.values())
    for n in gp_l:
        assert torch.allclose(gp_i[n], gp_l[n], rtol=5e-2, atol=5e-3 * float(gp_l[n].abs().max())), n
    assert torch.allclose(gx_i, gx_l, rtol=5e-2, atol=5e-3 * float(gx_l.abs().max()))


@pytest.mark.parametrize('shape', [(1024, 128, 512), (1024, 512, 128), (256, 512, 512), (4096, 4096, 128), (96, 160, 192)])
@pytest.mark.parametrize('with_res', [False, True])
@pytest.mark.parametrize('quantizers', ['all', 'no_dense', 'no_out'])
def test_linear_i8_nonorm_tail_equals_two_launches(shape, with_res, quantizers):
    """tq_linear_i8_nonorm_fwd (Linear -> (+ residual -> Q_sum) -> NoNorm -> Q_out in the GEMM epilogue) against the
    forms it replaces -- tq_linear_i8_fwd without output quantizer, then the quantizers / affine one kernel each, or
    tq_residual_nonorm_quant_fwd: bit-identical y and indices, on LDS-staged 64 / 128 tiles and the LDS-free kernel."""
    from quantization import _hip
    be = _hip.backend()
    M, N, K = shape
    p = _problem(M, N, K, 4, 4, False, seed=M + N + K + 7)
    dev = lambda t: t.cuda()
    x_i8 = be.quantize_to_int8(dev(p['x_q']), dev(p['xd']), dev(p['xz']), None, 4, False, False, 1e-8, 1, 1, minus_128=True)
    w_i8 = be.quantize_to_int8(dev(p['w_q']), dev(p['wd']), None, dev(torch.tensor(True)), 4, True, False, 1e-8, 1, 1,
                               minus_128=False)
    rs = be.rowsum_i8(w_i8)
    xq = (dev(p['xd']), dev(p['xz']), 4, 1e-8)
    wd = dev(p['wd']).reshape(-1)
    g = torch.Generator().manual_seed(5)
    res = (torch.randn(M, N, generator=g) * 0.8).cuda() if with_res else None
    nw = (1 + 0.3 * torch.randn(N, generator=g)).cuda()
    nb = (0.2 * torch.randn(N, generator=g)).cuda()
    lin = be.linear_i8(x_i8, w_i8, rs, dev(p['b']), xq, wd, 1e-8, _hip.ACT_NONE, None, torch.float32)

    def q7(lo, hi, bits):
        d, z = O.asym_params_from_range(lo, hi, bits)
        return (d.cuda(), z.cuda(), None, bits, False, False, 1e-8)
    s = float(lin.abs().max())
    q_dense = None if quantizers == 'no_dense' else q7(-0.8 * s, 0.7 * s, 4)
    q_sum = q7(-0.9 * s - 1, 0.9 * s + 1, 8) if with_res else None
    q_out = None if quantizers == 'no_out' else q7(-1.1 * s, 1.2 * s, 4)
    # the layered form: one kernel per quantizer / affine (each bit-exact against the oracle, tests/test_hip_parity.py)
    t = lin if q_dense is None else be.fake_quant(lin, *q_dense, 1, 1)[0]
    if with_res:
        t = be.fake_quant(t + res, *q_sum, 1, 1)[0]
    ref = be.affine_fake_quant(t, nw, nb, *q_out, want_idx=True) if q_out is not None else t * nw + nb
    if with_res and N in (128, 512):        # and the fused tail kernel of the two-launch form
        two = be.residual_layernorm_quant(lin, res, q_dense, q_sum, nw, nb, None, q_out)
        assert torch.equal(two, ref[0] if isinstance(ref, tuple) else ref)
    ref_y, ref_i = (ref if isinstance(ref, tuple) else (ref, None))
    got = be.linear_i8_nonorm(x_i8, w_i8, rs, dev(p['b']), res, nw, nb, xq, wd, 1e-8, q_dense, q_sum, q_out, torch.float32,
                              want_idx=q_out is not None)
    got_y, got_i = (got if isinstance(got, tuple) else (got, None))
    assert torch.equal(got_y, ref_y)
    if ref_i is not None:
        assert torch.equal(got_i, ref_i)
    yb = be.linear_i8_nonorm(x_i8, w_i8, rs, dev(p['b']), res, nw, nb, xq, wd, 1e-8, q_dense, q_sum, q_out, torch.bfloat16)
    assert torch.equal(yb, ref_y.to(torch.bfloat16))


@pytest.mark.parametrize('shape', [(1024, 128, 512), (64, 64, 128), (4096, 128, 512), (2048, 1024, 256)])
@pytest.mark.parametrize('quantizers', ['all', 'no_dense', 'no_out', 'all_plus_plain_linear'])
def test_grouped_linear_nonorm_pair_equals_two_launches(shape, quantizers):
    """tq_linear_i8_nonorm_grouped_fwd: two Linear -> NoNorm chains reading the same int8 input as one launch with
    split outputs (MobileBERT's two input bottlenecks) -- bit-identical to two tq_linear_i8_nonorm_fwd launches."""
    from quantization import _hip
    be = _hip.backend()
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    ps = [_problem(M, N, K, 8, 8, True, seed=7 * i + N) for i in range(2)]
    dev = lambda t: t.cuda()
    p0 = ps[0]
    x_i8 = be.quantize_to_int8(dev(p0['x_q']), dev(p0['xd']), dev(p0['xz']), None, 8, False, False, 1e-8, 1, 1, minus_128=True)
    xq = (dev(p0['xd']), dev(p0['xz']), 8, 1e-8)
    parts = []
    for p in ps:
        w_i8 = be.quantize_to_int8(dev(p['w_q']), dev(p['wd']), None, dev(torch.tensor(True)), 8, True, False, 1e-8, N, K,
                                   minus_128=False)
        nn_w = dev(torch.randn(N, generator=g) * 0.5 + 1.0)
        nn_b = dev(torch.randn(N, generator=g) * 0.2)
        pre = torch.nn.functional.linear(p0['x_q'], p['w_q'], p['b'])
        dd, dz = O.asym_params_from_range(pre.min(), pre.max(), 8)
        post = pre * nn_w.cpu() + nn_b.cpu()
        od, oz = O.asym_params_from_range(post.min(), post.max(), 8)
        parts.append(dict(w=w_i8, rs=be.rowsum_i8(w_i8), b=dev(p['b']), wd=dev(p['wd']).reshape(-1), nn_w=nn_w, nn_b=nn_b,
                          qd=(dev(dd), dev(dz), None, 8, False, False, 1e-8), qo=(dev(od), dev(oz), None, 8, False, False, 1e-8)))
    plain = None
    if quantizers == 'all_plus_plain_linear':
        # third group: a plain quantized Linear on the same input (MobileBERT's value Linear) as a chain with the identity
        # affine map and its own output quantizer twice -- must equal tq_linear_i8_fwd with that quantizer
        p = _problem(M, N, K, 8, 8, True, seed=99 + N)
        w_i8 = be.quantize_to_int8(dev(p['w_q']), dev(p['wd']), None, dev(torch.tensor(True)), 8, True, False, 1e-8, N, K,
                                   minus_128=False)
        pre = torch.nn.functional.linear(p0['x_q'], p['w_q'], p['b'])
        vd, vz = O.asym_params_from_range(pre.min(), pre.max(), 8)
        qv = (dev(vd), dev(vz), None, 8, False, False, 1e-8)
        plain = dict(w=w_i8, rs=be.rowsum_i8(w_i8), b=dev(p['b']), wd=dev(p['wd']).reshape(-1), nn_w=torch.ones(N).cuda(),
                     nn_b=torch.zeros(N).cuda(), qd=qv, qo=qv)
        parts.append(plain)
    qd = None if quantizers == 'no_dense' else [q['qd'] for q in parts]
    qo = None if quantizers == 'no_out' else [q['qo'] for q in parts]
    want_idx = qo is not None
    cat = lambda k: torch.cat([q[k] for q in parts]).contiguous()
    out = be.linear_i8_nonorm_grouped(x_i8, cat('w'), cat('rs'), cat('b'), cat('nn_w'), cat('nn_b'), xq, cat('wd'), 1e-8, qd,
                                      qo, torch.float32, want_idx=want_idx, n_groups=len(parts))
    ys, idxs = out if want_idx else (out, [None] * len(parts))
    if plain is not None:
        vy, vi = be.linear_i8(x_i8, plain['w'], plain['rs'], plain['b'], xq, plain['wd'], 1e-8, _hip.ACT_NONE, plain['qo'],
                              torch.float32, want_idx=True)
        assert torch.equal(ys[2], vy) and torch.equal(idxs[2], vi)
    for i, q in enumerate(parts):
        ref = be.linear_i8_nonorm(x_i8, q['w'], q['rs'], q['b'], None, q['nn_w'], q['nn_b'], xq, q['wd'], 1e-8,
                                  None if qd is None else q['qd'], None, None if qo is None else q['qo'], torch.float32,
                                  want_idx=want_idx)
        ry, ri = ref if want_idx else (ref, None)
        assert ys[i].is_contiguous() and torch.equal(ys[i], ry), i
        if want_idx:
            assert torch.equal(idxs[i], ri), i


@pytest.mark.parametrize('M', [32, 1024, 4096])
@pytest.mark.parametrize('quantizers', ['all', 'no_dense', 'no_sum', 'no_out'])
@pytest.mark.parametrize('per_channel', [False, True])
def test_ffn_i8_block_equals_two_launches(M, quantizers, per_channel):
    """tq_ffn_i8_nonorm_fwd -- MobileBERT's feed-forward block (128 -> 512 ReLU quant -> 128, residual NoNorm tail) as one
    launch with the intermediate kept in LDS -- against tq_linear_i8_fwd followed by tq_linear_i8_nonorm_fwd: bit-identical
    y and indices (same integer contractions, same element arithmetic)."""
    from quantization import _hip
    be = _hip.backend()
    K1, N1, N2 = 128, 512, 128
    g = torch.Generator().manual_seed(M + 17 * per_channel)
    x = torch.randn(M, K1, generator=g) * 1.2 + 0.2
    w1 = torch.randn(N1, K1, generator=g) * 0.08
    w2 = torch.randn(N2, N1, generator=g) * 0.06
    b1, b2 = torch.randn(N1, generator=g) * 0.1, torch.randn(N2, generator=g) * 0.1
    res = (torch.randn(M, N2, generator=g) * 0.7).cuda()
    nw, nb = (1 + 0.3 * torch.randn(N2, generator=g)).cuda(), (0.2 * torch.randn(N2, generator=g)).cuda()
    xd, xz = O.asym_params_from_range(x.min(), x.max(), 4)

    def wq(w, bits):
        d, _ = O.sym_params_from_range(w.amin(1) if per_channel else w.min(), w.amax(1) if per_channel else w.max(), bits)
        n = w.shape[0] if per_channel else 1
        wi = be.quantize_to_int8(w.cuda(), d.cuda(), None, torch.tensor(True).cuda(), bits, True, False, 1e-8, n,
                                 w.shape[1] if per_channel else 1, minus_128=False)
        return wi, be.rowsum_i8(wi), d.cuda().reshape(-1)
    w1i, rs1, w1d = wq(w1, 4)
    w2i, rs2, w2d = wq(w2, 4)
    x_i8 = be.quantize_to_int8(x.cuda(), xd.cuda(), xz.cuda(), None, 4, False, False, 1e-8, 1, 1, minus_128=True)
    xq = (xd.cuda(), xz.cuda(), 4, 1e-8)

    def q7(lo, hi, bits):
        d, z = O.asym_params_from_range(lo, hi, bits)
        return (d.cuda(), z.cuda(), None, bits, False, False, 1e-8)
    pre = be.linear_i8(x_i8, w1i, rs1, b1.cuda(), xq, w1d, 1e-8, _hip.ACT_RELU, None, torch.float32)
    q_mid = q7(0.0, 0.8 * float(pre.max()), 4)
    # two launches
    h, h_idx = be.linear_i8(x_i8, w1i, rs1, b1.cuda(), xq, w1d, 1e-8, _hip.ACT_RELU, q_mid, torch.float32, want_idx=True)
    mid_q = (q_mid[0], q_mid[1], 4, 1e-8)
    lin2 = be.linear_i8(h_idx, w2i, rs2, b2.cuda(), mid_q, w2d, 1e-8, _hip.ACT_NONE, None, torch.float32)
    s = float(lin2.abs().max())
    q_dense = None if quantizers == 'no_dense' else q7(-0.8 * s, 0.9 * s, 4)
    q_sum = None if quantizers == 'no_sum' else q7(-1.2 * s - 1, 1.1 * s + 1, 8)
    q_out = None if quantizers == 'no_out' else q7(-1.5 * s, 1.4 * s, 4)
    ref = be.linear_i8_nonorm(h_idx, w2i, rs2, b2.cuda(), res, nw, nb, mid_q, w2d, 1e-8, q_dense, q_sum, q_out,
                              torch.float32, want_idx=q_out is not None)
    got = be.ffn_i8_nonorm(x_i8, xq, w1i, rs1, b1.cuda(), w1d, 1e-8, q_mid, w2i, rs2, b2.cuda(), w2d, 1e-8, res, nw, nb,
                           q_dense, q_sum, q_out, torch.float32, want_idx=q_out is not None)
    ref_y, ref_i = ref if isinstance(ref, tuple) else (ref, None)
    got_y, got_i = got if isinstance(got, tuple) else (got, None)
    assert torch.equal(got_y, ref_y)
    if ref_i is not None:
        assert torch.equal(got_i, ref_i)
    yb = be.ffn_i8_nonorm(x_i8, xq, w1i, rs1, b1.cuda(), w1d, 1e-8, q_mid, w2i, rs2, b2.cuda(), w2d, 1e-8, res, nw, nb,
                          q_dense, q_sum, q_out, torch.bfloat16)
    assert torch.equal(yb, ref_y.to(torch.bfloat16))


@pytest.mark.parametrize('n_blocks', [2, 3, 4])
@pytest.mark.parametrize('M', [32, 1024])
@pytest.mark.parametrize('variant', ['all_fp32', 'mixed_fp32', 'all_bf16', 'slow_quantizer'])
@pytest.mark.parametrize('waves', [8, 4])
def test_ffn_chain_equals_consecutive_blocks(n_blocks, M, variant, waves, monkeypatch):
    """tq_ffn_chain_i8_nonorm_fwd -- consecutive MobileBERT feed-forward blocks in ONE launch, the rows staying on the CU
    (the output indices become the next x tile in LDS, y = scale * (index - zp) the next residual) -- against n launches of
    tq_ffn_i8_nonorm_fwd: bit-identical y and indices.  `mixed`: blocks without dense / sum quantizer, per-channel
    weight scales in some blocks; `slow_quantizer`: a scale outside the exact-quotient range sends one block through the
    division epilogue."""
    from quantization import _hip
    be = _hip.backend()
    monkeypatch.setenv('TQ_FFN_CHAIN_WAVES', str(waves))     # 8 (default) or 4 waves share the 16 rows of a workgroup
    K1, N1, N2 = 128, 512, 128
    g = torch.Generator().manual_seed(M + 31 * n_blocks)
    dt = torch.bfloat16 if variant == 'all_bf16' else torch.float32
    x = torch.randn(M, K1, generator=g) * 1.2 + 0.2
    xd, xz = O.asym_params_from_range(x.min(), x.max(), 8)
    x_i8 = be.quantize_to_int8(x.cuda(), xd.cuda(), xz.cuda(), None, 8, False, False, 1e-8, 1, 1, minus_128=True)
    xq = (xd.cuda(), xz.cuda(), 8, 1e-8)
    res = ((x_i8.float() + 128 - O.effective_zero_point(xz, 8).cuda()) * xd.cuda()).contiguous()     # the fp32 values of the input

    def q7(lo, hi, bits):
        d, z = O.asym_params_from_range(torch.tensor(float(lo)), torch.tensor(float(hi)), bits)
        return (d.cuda(), z.cuda(), None, bits, False, False, 1e-8)

    def wq(w, bits, per_channel):
        d, _ = O.sym_params_from_range(w.amin(1) if per_channel else w.min(), w.amax(1) if per_channel else w.max(), bits)
        n = w.shape[0] if per_channel else 1
        wi = be.quantize_to_int8(w.cuda(), d.cuda(), None, torch.tensor(True).cuda(), bits, True, False, 1e-8, n,
                                 w.shape[1] if per_channel else 1, minus_128=False)
        return wi, be.rowsum_i8(wi), d.cuda().reshape(-1)
    stages, cur_idx, cur_q, cur_res = [], x_i8, xq, res
    ref_y = ref_i = None
    for k in range(n_blocks):
        pc = variant == 'mixed_fp32' and k % 2 == 1
        w1i, rs1, w1d = wq(torch.randn(N1, K1, generator=g) * 0.08, 8, pc)
        w2i, rs2, w2d = wq(torch.randn(N2, N1, generator=g) * 0.06, 8, pc)
        b1 = (torch.randn(N1, generator=g) * 0.1).cuda()
        b2 = None if (variant == 'mixed_fp32' and k == 0) else (torch.randn(N2, generator=g) * 0.1).cuda()
        nw, nb = (1 + 0.3 * torch.randn(N2, generator=g)).cuda(), (0.2 * torch.randn(N2, generator=g)).cuda()
        pre = be.linear_i8(cur_idx, w1i, rs1, b1, cur_q, w1d, 1e-8, _hip.ACT_RELU, None, torch.float32)
        q_mid = q7(0.0, 0.8 * float(pre.max()), 8)
        _, h_idx = be.linear_i8(cur_idx, w1i, rs1, b1, cur_q, w1d, 1e-8, _hip.ACT_RELU, q_mid, torch.float32, want_idx=True)
        lin2 = be.linear_i8(h_idx, w2i, rs2, b2, (q_mid[0], q_mid[1], 8, 1e-8), w2d, 1e-8, _hip.ACT_NONE, None, torch.float32)
        s_ = float(lin2.abs().max()) + float(cur_res.float().abs().max())
        q_dense = None if (variant == 'mixed_fp32' and k == 1) else q7(-0.9 * s_, 0.9 * s_, 8)
        q_sum = None if (variant == 'mixed_fp32' and k == 2) else q7(-1.2 * s_, 1.1 * s_, 8)
        q_out = q7(-1.6 * s_, 1.5 * s_, 8)
        if variant == 'slow_quantizer' and k == 1:     # scale 2^110: outside the exact-quotient range -> division path
            q_sum = (torch.tensor(2.0 ** 110).cuda(), torch.tensor(128.0).cuda(), None, 8, False, False, 1e-8)
        st = dict(w1_idx=w1i, w1_rowsum=rs1, bias1=b1, w1_delta=w1d, w1_eps=1e-8, q_mid=q_mid, w2_idx=w2i, w2_rowsum=rs2,
                  bias2=b2, w2_delta=w2d, w2_eps=1e-8, nn_w=nw, nn_b=nb, q_dense=q_dense, q_sum=q_sum, q_out=q_out)
        stages.append(st)
        ref_y, ref_i = be.ffn_i8_nonorm(cur_idx, cur_q, w1i, rs1, b1, w1d, 1e-8, q_mid, w2i, rs2, b2, w2d, 1e-8, cur_res, nw, nb,
                                        q_dense, q_sum, q_out, dt, want_idx=True)
        cur_idx, cur_q, cur_res = ref_i, (q_out[0], q_out[1], 8, 1e-8), ref_y
    got_y, got_i = be.ffn_chain_i8_nonorm(x_i8, xq, res, stages, dt, want_idx=True)
    assert torch.equal(got_i, ref_i)
    assert torch.equal(got_y, ref_y)
    got_only_y = be.ffn_chain_i8_nonorm(x_i8, xq, res, stages, dt)
    assert torch.equal(got_only_y, ref_y)


@pytest.mark.gpu
def test_bert_ffn_with_index_only_intermediate_equals_separate_launches():
    """quantized_bert_ffn: the intermediate Linear (768 -> 3072, GELU, 8-bit quantizer) runs INDEX-ONLY
    (tq_linear_i8_fwd with y = NULL), the output Linear consumes the int8 indices, then the residual + LayerNorm tail:
    bit-identical to the separate calls (which store and never re-read the [B, T, 3072] fp32 tensor), layer by layer
    and for the whole BERT-base forward; and the producer really skipped its fp32 output."""
    from tests.test_bert_e2e import _build, _fixture, _calibrate_and_run
    from tests.harness_bert import QResidualBlock, QSelfAttention
    from harness.bert import QLayer
    from quantization import _hip, options
    z = _fixture()
    model, _ = _build('cuda')
    ids = torch.from_numpy(z['input_ids']).cuda()
    _calibrate_and_run(model, ids)
    be = _hip.backend()
    seen = []
    orig = be.linear_i8
    options.INT8_LINEAR = True
    QResidualBlock.fuse = QSelfAttention.fuse = True
    try:
        with torch.no_grad():
            separate = model(ids)
            be.linear_i8 = lambda *a, **k: (seen.append(k.get('want_y', True)), orig(*a, **k))[1]
            QLayer.fuse_ffn = True
            chained = model(ids)
    finally:
        QLayer.fuse_ffn = False
        QResidualBlock.fuse = QSelfAttention.fuse = False
        options.INT8_LINEAR = False
        be.__dict__.pop('linear_i8', None)
    assert seen.count(False) == 12, seen           # one index-only intermediate per encoder layer
    assert torch.equal(chained, separate)
