"""QAT on the whole (2-layer) BERT harness: utils.qat_utils.prepare_model_for_quantization, then a few
optimizer steps with (a) learnable ranges and (b) ranges estimated during training -- the STE backward
kernel, the range gradients and the train-mode estimators working together end to end."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(learn_ranges, fix_act=False):
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from tests.harness_bert import build_bert_base
    from utils.qat_utils import prepare_model_for_quantization
    from utils.utils import DotDict
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
              weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)
    model, _ = build_bert_base(seed=1000, num_layers=2, **qp)
    model = model.cuda()
    g = torch.Generator().manual_seed(0)
    batches = [(torch.randint(1000, 30000, (8, 64), generator=g).cuda(),) for _ in range(3)]
    labels = torch.randint(0, 2, (8,), generator=g).cuda()
    config = DotDict(quant=DotDict(act_quant=True, weight_quant=True),
                     act_quant=DotDict(num_batches=2, cross_entropy_layer=None),
                     qat=DotDict(learn_ranges=learn_ranges, fix_weight_ranges=False, fix_act_ranges=fix_act))
    prepare_model_for_quantization(config, model, batches)
    return model, batches, labels


def _train(model, batches, labels, steps=4, lr=1e-3):
    model.train()
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=lr)
    losses = []
    for i in range(steps):
        opt.zero_grad()
        logits = model(batches[i % len(batches)][0])
        loss = torch.nn.functional.cross_entropy(logits, labels)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses


def test_qat_with_learnable_ranges():
    model, batches, labels = _setup(learn_ranges=True)
    ranges = {n: p for n, p in model.named_parameters() if n.endswith('_delta') or n.endswith('_zero_float')}
    assert len(ranges) > 50, 'learn_ranges must turn the quantizer buffers into parameters'
    before = {n: p.detach().clone() for n, p in ranges.items()}
    losses = _train(model, batches, labels)
    assert all(torch.isfinite(torch.tensor(losses))), losses
    with_grad = [n for n, p in ranges.items() if p.grad is not None and torch.isfinite(p.grad).all()]
    assert len(with_grad) >= 0.9 * len(ranges)
    moved = sum(int(not torch.equal(before[n], p.detach())) for n, p in ranges.items())
    assert moved > len(ranges) // 2, 'range parameters must receive STE gradients and be updated'


@pytest.mark.parametrize('fix_act', [False, True])
def test_qat_with_estimated_ranges(fix_act):
    from quantization.quantization_manager import QuantizationManager, Qstates
    model, batches, labels = _setup(learn_ranges=False, fix_act=fix_act)
    acts = [m for n, m in model.named_modules() if isinstance(m, QuantizationManager) and n.endswith('activation_quantizer')]
    want = Qstates.fix_ranges if fix_act else Qstates.estimate_ranges_train
    assert all(m.state == want for m in acts)
    snap = [m.quantizer._delta.detach().clone() for m in acts]
    losses = _train(model, batches, labels)
    assert all(torch.isfinite(torch.tensor(losses))), losses
    changed = sum(int(not torch.equal(a, m.quantizer._delta.detach().reshape(a.shape))) for a, m in zip(snap, acts))
    assert (changed == 0) if fix_act else (changed > len(acts) // 2)
    model.eval()
    with torch.no_grad():
        out = model(batches[0][0])
        d0 = [m.quantizer._delta.detach().clone() for m in acts]
        model(batches[1][0])                       # eval mode: estimate_ranges_train must not move the ranges
    assert all(torch.equal(a, m.quantizer._delta.detach()) for a, m in zip(d0, acts)) and torch.isfinite(out).all()


@pytest.mark.parametrize('layout,symmetric', [('per_channel', True), ('per_channel', False), ('per_embd', False),
                                              ('axis1', False)])   # per-axis needs an asymmetric quantizer (quirk q3)
def test_learnable_vector_ranges_gradients_vs_autograd(layout, symmetric):
    """`learn_ranges()` on per-channel weights / per-embedding activations: d loss / d _delta and d loss / d _zero_float
    per parameter (tq_fake_quant_bwd with n_params > 1) against plain autograd through the reference's op chain with
    the STE round (reference quantizers.py:12-19, 142-153, 184-185, 209).  Tolerance 1e-4 relative (fp32 sums)."""
    from oracle import tq_oracle as O
    from quantization.quantizers import AsymmetricUniformQuantizer, SymmetricUniformQuantizer
    g = torch.Generator().manual_seed(3)
    if layout == 'per_channel':
        x = torch.randn(48, 96, generator=g) * torch.linspace(0.5, 3, 48)[:, None]
        axis, per_channel, pshape = None, True, (48, 1)
        mn, mx = x.min(1)[0], x.max(1)[0]
    elif layout == 'per_embd':
        x = torch.randn(4, 16, 96, generator=g) * torch.linspace(0.5, 3, 96)
        axis, per_channel, pshape = 2, False, (1, 1, 96)
        mn, mx = x.reshape(-1, 96).min(0)[0], x.reshape(-1, 96).max(0)[0]
    else:
        x = torch.randn(4, 16, 96, generator=g)
        axis, per_channel, pshape = 1, False, (1, 16, 1)
        mn, mx = x.permute(1, 0, 2).reshape(16, -1).min(1)[0], x.permute(1, 0, 2).reshape(16, -1).max(1)[0]
    mn, mx = mn * 0.7, mx * 0.7                       # clip: both gradient branches are exercised
    cls = SymmetricUniformQuantizer if symmetric else AsymmetricUniformQuantizer
    q = cls(n_bits=4, per_channel=per_channel, axis=axis).cuda()
    q.set_quant_range(mn.cuda(), mx.cuda())
    xd = x.cuda().requires_grad_(True)
    with torch.no_grad():
        q(xd)                                          # shapes the parameter views ([C,1] / [1,1,d])
    q.make_range_trainable()
    gy = torch.randn(x.shape, generator=g)
    y = q(xd)
    y.backward(gy.cuda())
    delta = q._delta.detach().cpu().reshape(pshape)
    zf = None if symmetric else q._zero_float.detach().cpu().reshape(pshape)
    signed = bool(q._signed.item()) if symmetric else False
    _, dx, dd, dz = O.fake_quant_with_grads(x, delta, zf, 4, symmetric, signed, grad_out=gy)
    assert torch.allclose(xd.grad.cpu(), dx, rtol=1e-5, atol=1e-6)
    assert q._delta.grad is not None and q._delta.grad.shape == q._delta.shape
    assert torch.allclose(q._delta.grad.cpu().reshape(pshape), dd, rtol=1e-4, atol=1e-4), layout
    if not symmetric:
        assert torch.allclose(q._zero_float.grad.cpu().reshape(pshape), dz, rtol=1e-4, atol=1e-4), layout


@pytest.mark.parametrize('optimizer,int8', [('sgd', False), ('adam', False), ('sgd', True)])
def test_graphed_qat_step_equals_eager(optimizer, int8):
    """quantization.graphs.GraphedTrainStep: zero_grad + forward + loss + STE backward + optimizer.step() as ONE
    hipGraph.  Same parameters (weights AND learnable ranges) after N replays as after N eager steps from the same
    state: the backward kernels reduce range gradients in a fixed order, dropout is switched off, rocBLAS is
    deterministic for a fixed shape."""
    import copy
    from quantization import options
    from quantization.autoquant_utils import INT8_STATS
    from quantization.graphs import GraphedTrainStep
    # int8: the Linears run on the i8 matrix cores under autograd (_Int8LinearSTE); that path needs fixed ranges
    model, batches, labels = _setup(learn_ranges=not int8, fix_act=int8)
    if int8:
        model.fix_ranges()
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    twin = copy.deepcopy(model)

    def make_opt(net):
        params = [p for p in net.parameters() if p.requires_grad]
        if optimizer == 'sgd':
            return torch.optim.SGD(params, lr=1e-3, momentum=0.9)
        return torch.optim.Adam(params, lr=1e-4, capturable=True)

    loss_fn = torch.nn.functional.cross_entropy
    opt = make_opt(model)
    options.INT8_LINEAR = int8
    try:
        _graphed_vs_eager(model, twin, make_opt, opt, loss_fn, batches, labels, int8, INT8_STATS, GraphedTrainStep)
    finally:
        options.INT8_LINEAR = False


def _graphed_vs_eager(model, twin, make_opt, opt, loss_fn, batches, labels, int8, INT8_STATS, GraphedTrainStep):
    before = INT8_STATS['autograd_calls']
    step = GraphedTrainStep(model, loss_fn, opt, (batches[0][0],), (labels,))
    assert (INT8_STATS['autograd_calls'] > before) == int8
    # capture (with its warm-up steps) left the model where it was
    for (n, a), (_, b) in zip(model.state_dict().items(), twin.state_dict().items()):
        assert torch.equal(a, b), n
    graph_losses = []
    for i in range(4):
        graph_losses.append(float(step((batches[i % 3][0],), (labels,)).clone()))

    opt2 = make_opt(twin)
    eager_losses = []
    for i in range(4):
        opt2.zero_grad(set_to_none=True)
        loss = loss_fn(twin(batches[i % 3][0]), labels)
        loss.backward()
        opt2.step()
        eager_losses.append(float(loss.detach()))
    assert graph_losses == eager_losses, (graph_losses, eager_losses)
    moved = 0
    for (n, a), (_, b) in zip(model.named_parameters(), twin.named_parameters()):
        assert torch.equal(a.detach(), b.detach()), n
        moved += int(a.requires_grad)
    assert moved > (30 if int8 else 100)


def test_graph_replays_invalidate_derived_caches_and_keep_optimizer_state():
    """a hipGraph replay rewrites weights in place without bumping tensor._version; caches keyed on the
    version (int8 weight indices of the integer Linears, NoNorm parameters, stacked operands) must not survive it.
    Sequence: graphed QAT steps -> eval (fills the caches) -> more replays -> eval again: the second eval must see the
    NEW weights (== an eval with every cache dropped by hand), and must differ from the first.  Also: capturing with an
    optimizer that has already stepped keeps its moments / step count (they are snapshot and restored, not zeroed)."""
    from quantization import options
    from quantization.autoquant_utils import QuantLinear
    from quantization.graphs import GraphedTrainStep
    model, batches, labels = _setup(learn_ranges=False, fix_act=True)
    model.fix_ranges()
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    params = [p for p in model.parameters() if p.requires_grad]
    # (lr: large enough that three replays move the 8-bit logits, small enough that the 2-layer network does not run into
    # the clamp of its output quantizer -- at 5e-2 it did with the round-4 weights and every logit became the same value)
    opt = torch.optim.SGD(params, lr=1e-2, momentum=0.9)
    loss_fn = torch.nn.functional.cross_entropy
    # an optimizer with history: two eager steps first
    for i in range(2):
        opt.zero_grad(set_to_none=True)
        loss_fn(model(batches[i][0]), labels).backward()
        opt.step()
    mom_before = {id(p): opt.state[p]['momentum_buffer'].clone() for p in params if p in opt.state}
    assert mom_before and any(float(v.abs().max()) > 0 for v in mom_before.values())
    options.INT8_LINEAR = True
    try:
        step = GraphedTrainStep(model, loss_fn, opt, (batches[0][0],), (labels,))
        for p in params:                                           # restored, not zeroed
            if id(p) in mom_before:
                assert torch.equal(opt.state[p]['momentum_buffer'], mom_before[id(p)])
        step((batches[0][0],), (labels,))
        model.eval()
        with torch.no_grad():
            out1 = model(batches[1][0]).clone()                    # fills the int8 weight caches
        assert any(m._int8_cache is not None for m in model.modules() if isinstance(m, QuantLinear))
        model.train()
        for i in range(3):
            step((batches[i % 3][0],), (labels,))
        model.eval()
        with torch.no_grad():
            out2 = model(batches[1][0]).clone()
            for m in model.modules():                              # ground truth: every derived cache dropped by hand
                if isinstance(m, QuantLinear):
                    m._int8_cache = None
            out3 = model(batches[1][0]).clone()
    finally:
        options.INT8_LINEAR = False
    assert torch.equal(out2, out3)
    assert not torch.equal(out2, out1)
