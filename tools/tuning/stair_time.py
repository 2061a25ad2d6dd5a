"""Linear + GELU + 8-bit quantizer on the i8 matrix cores: arithmetic epilogue vs the staircase table (csrc/tq_stair.hip).
hipGraph of 20 back-to-back launches, fp32 y + int8 indices and index-only."""
import sys
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization import _hip
be = _hip.backend(); dev = 'cuda'


def graph_time(fn, n=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 5 / n * 1e3


for (M, N, K) in [(1024, 3072, 768), (8192, 3072, 768), (16384, 3072, 768)]:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randint(-128, 127, (M, K), dtype=torch.int8, device=dev, generator=g)
    w = torch.randint(-127, 127, (N, K), dtype=torch.int8, device=dev, generator=g)
    rs = be.rowsum_i8(w); b = torch.randn(N, device=dev)
    xq = (torch.tensor(0.02, device=dev), torch.tensor(117.0, device=dev), 8, 1e-8)
    wd = torch.tensor(0.0004, device=dev).reshape(1)
    q = (torch.tensor(0.036, device=dev), torch.tensor(5.0, device=dev), None, 8, False, False, 1e-8)
    stair = be.act_stair(_hip.ACT_GELU, q)
    ok = stair[0][:16].view(torch.float32).cpu().tolist()
    y0, i0 = be.linear_i8(x, w, rs, b, xq, wd, 1e-8, _hip.ACT_GELU, q, torch.float32, want_idx=True)
    y1, i1 = be.linear_i8(x, w, rs, b, xq, wd, 1e-8, _hip.ACT_GELU, q, torch.float32, want_idx=True, stair=stair)
    diff = (i0.int() - i1.int()).abs()
    res = {}
    for name, st in (('arith', None), ('stair', stair)):
        res[name] = (graph_time(lambda: be.linear_i8(x, w, rs, b, xq, wd, 1e-8, _hip.ACT_GELU, q, torch.float32, want_idx=True, stair=st)),
                     graph_time(lambda: be.linear_i8(x, w, rs, b, xq, wd, 1e-8, _hip.ACT_GELU, q, torch.float32, want_idx=True, want_y=False, stair=st)))
    ops = 2 * M * N * K
    print(f'M={M} N={N} K={K} header={ok} levels={i0.unique().numel()} differing={int((diff != 0).sum())} maxdiff={int(diff.max())}')
    for name, (t, ti) in res.items():
        print(f'   {name}: y+idx {t:7.2f} us ({ops / t / 1e6 / 3944 * 100:5.1f} % of 3.944 POP/s)   index-only {ti:7.2f} us ({ops / ti / 1e6 / 3944 * 100:5.1f} %)', flush=True)
