"""Per-embedding fake-quant throughput; run with TQ_AXIS_REG=0 / 1 to compare the LDS-table and the
register-resident kernels.  Also checks bit-equality against the scalar kernel result."""
import os, sys, time
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization import _hip
be = _hip.backend()
dev = 'cuda'
def ev(fn, n=10, w=3, rounds=5):
    for _ in range(w): fn()
    best = 1e9
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n)
    return best
t0 = time.time()
x = torch.empty(1 << 28, device=dev)
while time.time() - t0 < 0.5: x.mul_(1.0)      # clock settle
del x
print('TQ_AXIS_REG =', os.environ.get('TQ_AXIS_REG', '(default)'), 'TQ_AXIS_TPB =', os.environ.get('TQ_AXIS_TPB', '(auto)'))
ONLY = os.environ.get('ONLY_BF16')
for rows, d in ((1024 * 512, 768), (1024 * 512, 1024), (1024 * 128, 3072), (1024 * 512, 128), (8 * 128, 768), (1024 * 512, 520)):
    for dt in ((torch.bfloat16,) if ONLY else (torch.bfloat16, torch.float32)):
        x = (torch.randn(rows, d, device=dev) * 3).to(dt)
        delta = torch.rand(d, device=dev) * 0.05 + 0.01
        zf = torch.rand(d, device=dev) * 255
        f = lambda: be.fake_quant(x, delta, zf, None, 8, False, False, 1e-8, d, 1)
        ms = ev(f)
        y = f()
        y = y[0] if isinstance(y, tuple) else y
        # reference: same op on an unaligned (scalar-kernel) view of a small slice
        xs = x[:64].contiguous()
        buf = torch.empty(xs.numel() + 1, device=dev, dtype=dt)[1:].view_as(xs); buf.copy_(xs)
        r = be.fake_quant(buf, delta, zf, None, 8, False, False, 1e-8, d, 1)
        r = r[0] if isinstance(r, tuple) else r
        ok = torch.equal(r, y[:64])
        print(f'[{rows},{d}] {str(dt)[6:]:9s} {ms*1e3:9.1f} us  {2*x.numel()*x.element_size()/ms/1e9:7.2f} TB/s  scalar-equal={ok}')
