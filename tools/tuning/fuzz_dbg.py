import sys
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import numpy as np, torch
from oracle import tq_oracle as O
from quantization import _hip
from tests.test_fuzz_parity import _case
from quantization.quantizers import param_layout
be = _hip.backend()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rs = np.random.RandomState(1000 + seed); g = torch.Generator().manual_seed(seed)
for it in range(60):
    layout, dtype, shape, axis, per_channel, sym, n_bits = _case(rs)
    x = (torch.randn(*shape, generator=g) * float(rs.choice([0.1, 1.0, 30.0]))).to(dtype)
    if rs.rand() < 0.2 and x.numel() > 3:
        x.view(-1)[rs.randint(x.numel())] = float('inf')
    n_par = shape[axis] if axis is not None else (shape[0] if per_channel else 1)
    lo = -np.abs(rs.randn(n_par)).astype(np.float32) * 2 - 0.01
    hi = np.abs(rs.randn(n_par)).astype(np.float32) * 2 + 0.01
    if n_par == 1: lo, hi = lo[0], hi[0]
    if sym:
        delta, signed = O.sym_params_from_range(torch.as_tensor(lo), torch.as_tensor(hi), n_bits); zf, sgn = None, bool(signed)
    else:
        delta, zf = O.asym_params_from_range(torch.as_tensor(lo), torch.as_tensor(hi), n_bits); signed, sgn = None, False
    ref_idx, ref_y = O.fake_quant_lowp(x, delta, zf, n_bits, sym, sgn, axis=axis, per_channel=per_channel)
    n_params, inner = param_layout(x, int(delta.numel()), axis, per_channel, tuple(delta.shape))
    xd = x.cuda()
    un = rs.rand() < 0.25 and x.numel() > 1
    if un:
        buf = torch.empty(x.numel() + 1, dtype=dtype, device='cuda'); xd = buf[1:].view(shape); xd.copy_(x)
    y, idx = be.fake_quant(xd, delta.reshape(-1).cuda(), None if zf is None else zf.reshape(-1).cuda(),
                           None if signed is None else signed.cuda(), n_bits, sym, False, 1e-8, n_params, inner, idx_dtype=torch.int32)
    fin = torch.isfinite(ref_idx)
    bad = (y.cpu() != ref_y) & fin
    if bad.any() or not torch.equal(idx.cpu().float()[fin], ref_idx[fin]):
        w = bad.nonzero()[:5]
        print(it, layout, dtype, shape, sym, n_bits, 'unaligned' if un else '', 'n_bad', int(bad.sum()))
        for i in w:
            i = tuple(int(v) for v in i)
            col = i[axis] if axis is not None else 0
            print('   x', float(x[i]), 'got', float(y.cpu()[i]), 'ref', float(ref_y[i]), 'idx', int(idx.cpu()[i]), float(ref_idx[i]),
                  'delta', float(delta.reshape(-1)[col]), 'zf', None if zf is None else float(zf.reshape(-1)[col]))
        break
