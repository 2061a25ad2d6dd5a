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


@pytest.mark.parametrize('seed', range(6))
def test_fake_quant_fuzz(seed):
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
