"""Breakdown of the fused Linear + GELU + quantizer kernel (index-only, M = 8192) with the -DTQ_I8_DBG_BUILD library:
TQ_I8_DBG bit 1 = no epilogue, 2 = no operand loads, 4 = no MFMA.  Run once per setting (the library reads the env per launch)."""
import os, sys
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization import _hip
be = _hip.backend(); dev = 'cuda'
M, N, K = 8192, 3072, 768


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


g = torch.Generator(device=dev).manual_seed(1)
x = torch.randint(-128, 127, (M, K), dtype=torch.int8, device=dev, generator=g)
w = torch.randint(-127, 127, (N, K), dtype=torch.int8, device=dev, generator=g)
rs = be.rowsum_i8(w); b = torch.randn(N, device=dev)
xq = (torch.tensor(0.02, device=dev), torch.tensor(117.0, device=dev), 8, 1e-8)
wd = torch.tensor(0.0004, device=dev).reshape(1)
q = (torch.tensor(0.036, device=dev), torch.tensor(5.0, device=dev), None, 8, False, False, 1e-8)
stair = be.act_stair(_hip.ACT_GELU, q)
for dbg in (0, 1, 2, 4, 3, 5, 6):
    os.environ['TQ_I8_DBG'] = str(dbg)
    t = {}
    for name, st in (('arith', None), ('stair', stair)):
        t[name] = graph_time(lambda: be.linear_i8(x, w, rs, b, xq, wd, 1e-8, _hip.ACT_GELU, q, torch.float32, want_idx=True, want_y=False, stair=st))
    t['quant-only'] = graph_time(lambda: be.linear_i8(x, w, rs, b, xq, wd, 1e-8, _hip.ACT_NONE, q, torch.float32, want_idx=True, want_y=False))
    print(f'TQ_I8_DBG={dbg} (1 no epilogue, 2 no loads, 4 no MFMA): ' + '  '.join(f'{k} {v:6.2f} us' for k, v in t.items()), flush=True)
