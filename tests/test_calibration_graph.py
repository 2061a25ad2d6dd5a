"""options.INPLACE_CALIBRATION_STATE: the fused calibration step updates estimator state and quantizer
parameters in place, so a calibrating forward (statistics -> EMA -> parameters -> quantize at every
quantizer, no host synchronisation) can be captured once and replayed as a hipGraph for every further
batch.  Bar: state_dict and outputs bit-identical to eager calibration over the same batches."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(num_layers, per_groups=None):
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from tests.harness_bert import build_bert_base, apply_activation_granularity
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
              weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)
    model, _ = build_bert_base(seed=1000, num_layers=num_layers, **qp)
    if per_groups:
        apply_activation_granularity(model, per_groups=per_groups)
    model = model.cuda().eval()
    model.set_quant_state(True, True)
    return model


def _batches(n, B=8, T=128):
    g = torch.Generator().manual_seed(3)
    return [torch.randint(1000, 30000, (B, T), generator=g).cuda() for _ in range(n)]


@pytest.mark.parametrize('per_groups', [None, 6], ids=['per-tensor', 'peg6'])
def test_calibrating_forward_replays_as_hipgraph(per_groups):
    from quantization import options
    batches = _batches(4)
    with torch.no_grad():
        ref = _model(2, per_groups)
        for b in batches:
            ref_out = ref(b)
        ref_sd = {k: v.clone() for k, v in ref.state_dict().items()}

        options.INPLACE_CALIBRATION_STATE = True
        try:
            m = _model(2, per_groups)
            m(batches[0])                                   # first batch eager: allocates every state buffer
            ptrs = {k: v.data_ptr() for k, v in m.state_dict().items()}
            snap = {k: v.clone() for k, v in m.state_dict().items()}
            static = batches[1].clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    m(static)                               # warm-up (workspaces, ticket words)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = m(static)
            sd = m.state_dict()
            assert {k: v.data_ptr() for k, v in sd.items()} == ptrs, 'state must stay in the same buffers'
            for k, v in sd.items():                         # undo the warm-up updates
                v.copy_(snap[k])
            for b in batches[1:]:
                static.copy_(b)
                g.replay()
            torch.cuda.synchronize()
        finally:
            options.INPLACE_CALIBRATION_STATE = False
    sd = m.state_dict()
    assert sd.keys() == ref_sd.keys()
    bad = [k for k in sd if not torch.equal(sd[k].reshape(-1), ref_sd[k].reshape(-1))]
    assert not bad, bad[:5]
    assert torch.equal(out, ref_out)


def test_inplace_state_matches_rebinding_eagerly():
    from quantization import options
    batches = _batches(3)
    with torch.no_grad():
        a = _model(1)
        outs_a = [a(b) for b in batches]
        options.INPLACE_CALIBRATION_STATE = True
        try:
            b_ = _model(1)
            outs_b = [b_(b) for b in batches]
        finally:
            options.INPLACE_CALIBRATION_STATE = False
    for x, y in zip(outs_a, outs_b):
        assert torch.equal(x, y)
    for (k, v), (k2, v2) in zip(a.state_dict().items(), b_.state_dict().items()):
        assert k == k2 and torch.equal(v.reshape(-1), v2.reshape(-1)), k


def test_graphed_forward_helper_calibrating_and_fixed():
    """quantization.graphs.GraphedForward: capture + state restore + replay == eager, for a calibrating forward
    (in-place state) and then for the fixed-range forward of the same model."""
    from quantization import options
    from quantization.graphs import GraphedForward
    batches = _batches(4)
    with torch.no_grad():
        ref = _model(2)
        for b in batches:
            ref(b)
        ref.fix_ranges()
        ref_fixed = ref(batches[0])
        options.INPLACE_CALIBRATION_STATE = True
        try:
            m = _model(2)
            m(batches[0])
            g = GraphedForward(m, batches[1])
            for b in batches[1:]:
                g(b)
        finally:
            options.INPLACE_CALIBRATION_STATE = False
        m.fix_ranges()
        gf = GraphedForward(m, batches[0])
        out = gf(batches[0]).clone()
        out2 = gf(batches[2]).clone()
        eager2 = m(batches[2])
    for (k, v), (k2, v2) in zip(m.state_dict().items(), ref.state_dict().items()):
        assert k == k2 and torch.equal(v.reshape(-1), v2.reshape(-1)), k
    assert torch.equal(out, ref_fixed) and torch.equal(out2, eager2)
    with pytest.raises(ValueError):
        gf(batches[0][:4])


def test_replay_survives_derived_caches_rebuilt_after_capture():
    """A recorded fixed-range forward with integer Linears reads its derived caches (cached quantized parameters, int8
    weight indices, row sums, GELU staircase tables) by ADDRESS.  When they are rebuilt afterwards -- here:
    `invalidate_derived_caches()` + an eager forward -- the old tensors would return to the allocator and be overwritten
    by the next allocation; the graph object keeps them alive, so the replay is unchanged."""
    from quantization import options
    from quantization.graphs import GraphedForward, derived_cache_tensors
    batches = _batches(3)
    options.INT8_LINEAR = True
    try:
        with torch.no_grad():
            m = _model(2)
            for b in batches:
                m(b)
            m.fix_ranges()
            g = GraphedForward(m, batches[0])
            want = g(batches[1]).clone()
            held = {t.data_ptr() for t in g._cache_refs}
            assert len(held) >= 2 * 6 * 2                    # per layer: six Linears x (indices, row sums) at least
            options.invalidate_derived_caches()
            m(batches[2])                                    # every derived cache is rebuilt: new tensors
            fresh = {t.data_ptr() for t in derived_cache_tensors(m)}
            assert len(held - fresh) >= 2 * 6 * 2          # int8 indices / row sums / tables were replaced (the reference's own
            #                                               cached_params are not keyed by the epoch and stay)
            junk = [torch.full((1 << 20,), 7, dtype=torch.int8, device='cuda') for _ in range(64)]   # recycle freed blocks
            torch.cuda.synchronize()
            got = g(batches[1]).clone()
            del junk
    finally:
        options.INT8_LINEAR = False
    assert torch.equal(got, want)
