"""tq_fake_quant_multi_fwd: many independent tensors, one launch (40 per launch) -- bit-identical to one
tq_fake_quant_fwd per tensor; and the model-level user, `prequantize_weights`, which fills the eval-mode parameter
cache of every layer (reference hijacker.py:52-64) in one go."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _items(dtype, seed, count):
    g = torch.Generator().manual_seed(seed)
    items = []
    for k in range(count):
        kind = k % 4
        if kind == 0:       # per-tensor asymmetric, ragged size
            x = torch.randn(int(torch.randint(1, 70000, (1,), generator=g)), generator=g) * 2
            q = (torch.tensor(0.013 + 0.001 * k), torch.tensor(float(100 + k)), None, 8, False, False, 1e-8, 1, 1)
        elif kind == 1:     # per-output-channel symmetric signed weights [N, K]
            N, K = int(torch.randint(1, 60, (1,), generator=g)), 8 * int(torch.randint(1, 40, (1,), generator=g))
            x = torch.randn(N, K, generator=g) * 0.05
            q = (torch.rand(N, generator=g) * 0.002 + 1e-4, None, torch.tensor(True), 4 + k % 5, True, False, 1e-8, N, K)
        elif kind == 2:     # per-tensor symmetric unsigned, 4 bits, with NaN / Inf / -0
            x = torch.rand(4099, generator=g) * 3
            x[::513] = float('nan'); x[1::1025] = float('inf'); x[2] = -0.0
            q = (torch.tensor(0.2), None, torch.tensor(False), 4, True, False, 1e-8, 1, 1)
        else:               # log-domain scale, 16-bit grid
            x = torch.randn(3, 5, 64, generator=g)
            q = (torch.tensor(-7.0), torch.tensor(31000.0), None, 16, False, True, 1e-8, 1, 1)
        items.append((x.to(dtype).to(DEV),) + tuple(t.to(DEV) if torch.is_tensor(t) else t for t in q))
    return items


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_multi_tensor_launch_equals_one_launch_per_tensor(dtype):
    from quantization import _hip
    be = _hip.backend()
    items = _items(dtype, 3, 97)                                  # > 2 x 40: split over three launches
    empty = (torch.empty(0, dtype=dtype, device=DEV),) + items[0][1:]
    items.insert(5, empty)
    ys = be.fake_quant_multi(items)
    assert len(ys) == len(items) and ys[5].numel() == 0
    for k, (it, y) in enumerate(zip(items, ys)):
        if it[0].numel() == 0:
            continue
        ref, _ = be.fake_quant(*it)
        same = (y == ref) | (torch.isnan(y) & torch.isnan(ref))
        assert y.shape == ref.shape and bool(same.all()), (k, it[0].shape)
    assert be.fake_quant_multi([]) == []


def test_multi_tensor_launch_rejects_what_it_cannot_tile():
    from quantization import _hip
    be = _hip.backend()
    x = torch.randn(6, 10, device=DEV)                            # rows of 10 floats: not whole 16-byte vectors
    bad = (x, torch.full((6,), 0.01, device=DEV), None, torch.tensor(True, device=DEV), 8, True, False, 1e-8, 6, 10)
    with pytest.raises(_hip.TQError, match='16-byte vectors'):
        be.fake_quant_multi([bad])
    with pytest.raises(_hip.TQError, match='one dtype'):
        be.fake_quant_multi([_items(torch.float32, 1, 1)[0], _items(torch.bfloat16, 1, 1)[0]])


@pytest.mark.layered_route
def test_prequantize_weights_fills_the_same_cache_as_the_lazy_path():
    from quantization.autoquant_utils import prequantize_weights
    from quantization.hijacker import QuantizationHijacker
    from tests.test_calibration_graph import _model, _batches
    batches = _batches(2)
    with torch.no_grad():
        lazy, pre = _model(2), _model(2)
        for m in (lazy, pre):
            m(batches[0])
            m.fix_ranges()
            for layer in m.modules():                             # the calibrating forward cached under the OLD state
                if isinstance(layer, QuantizationHijacker):
                    layer.cached_params = None
        calls = {'n': 0}
        from quantization import _hip
        be = _hip.backend()
        orig = be.lib.tq_fake_quant_multi_fwd

        class _Counting:
            def __call__(self, *a):
                calls['n'] += 1
                return orig(*a)
        be.lib.tq_fake_quant_multi_fwd = _Counting()
        try:
            served = prequantize_weights(pre)
        finally:
            be.lib.tq_fake_quant_multi_fwd = orig
        layers = [l for l in pre.modules() if isinstance(l, QuantizationHijacker) and l._quant_w]
        assert served == len(layers) >= 2 * 6 + 3 and calls['n'] == 1          # one C call (<= 40 tensors per launch inside)
        assert prequantize_weights(pre) == 0                       # everything is cached now
        out_lazy, out_pre = lazy(batches[1]), pre(batches[1])
        assert torch.equal(out_lazy, out_pre)
        for (n1, a), (n2, b) in zip(lazy.named_modules(), pre.named_modules()):
            if isinstance(a, QuantizationHijacker) and a._quant_w:
                assert a.cached_params is not None and b.cached_params is not None, n1
                assert torch.equal(a.cached_params[0], b.cached_params[0]), n1
                assert (a.cached_params[1] is None) == (b.cached_params[1] is None)
                if a.cached_params[1] is not None:
                    assert torch.equal(a.cached_params[1], b.cached_params[1]), n1
        pre.train()
        assert prequantize_weights(pre) == 0                       # training mode: nothing is cached
