"""Integer Linear at serving batch sizes: 128 x 128 block tiles against 64 x 64 (TQ_I8_BIG_MIN = 1 / huge), the four Linear
shapes of a BERT-base layer at M = 2048 ... 16384 tokens, as the forward runs them (pre-quantizer fp32 output for the two
Linears in front of a LayerNorm tail, int8 indices only for QKV and the GELU Linear); 30 calls in one hipGraph."""
import os, subprocess, sys
if len(sys.argv) == 1:
    for bm in ('1', '1000000'):
        print(f'== TQ_I8_BIG_MIN={bm} ({"128 x 128 tiles" if bm == "1" else "64 x 64 tiles"})', flush=True)
        subprocess.check_call([sys.executable, __file__, 'run'], env=dict(os.environ, TQ_I8_BIG_MIN=bm))
    sys.exit(0)
sys.path.insert(0, '/root/repo/transformer-quantization_amd')
import torch
from quantization import _hip
be = _hip.backend()
dev = 'cuda'
def q7(d, z):
    return (torch.tensor(d, device=dev), torch.tensor(z, device=dev), None, 8, False, False, 1e-8)
def graph_us(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1e3
for M in (2048, 4096, 8192, 16384, 32768):
    for (N, K, kind) in ((768, 768, 'attention output (fp32 out)'), (768, 3072, 'FFN2 (fp32 out)'), (2304, 768, 'QKV (index only)'),
                         (3072, 768, 'FFN1 GELU (index only)')):
        x = torch.randint(-128, 127, (M, K), dtype=torch.int8, device=dev)
        w = torch.randint(-127, 127, (N, K), dtype=torch.int8, device=dev)
        rs = be.rowsum_i8(w)
        b = torch.randn(N, device=dev)
        xd, xz = torch.tensor(0.02, device=dev), torch.tensor(117.0, device=dev)
        wd = torch.tensor(0.001, device=dev).reshape(1)
        qo = q7(0.05, 100.0)
        if 'fp32' in kind:
            fn = lambda: be.linear_i8(x, w, rs, b, (xd, xz, 8, 1e-8), wd, 1e-8, _hip.ACT_NONE, None, torch.float32)
        elif 'GELU' in kind:
            st = be.act_stair(_hip.ACT_GELU, qo)
            fn = lambda: be.linear_i8(x, w, rs, b, (xd, xz, 8, 1e-8), wd, 1e-8, _hip.ACT_GELU, qo, torch.float32, want_idx=True, want_y=False, stair=st)
        else:
            fn = lambda: be.linear_i8(x, w, rs, b, (xd, xz, 8, 1e-8), wd, 1e-8, _hip.ACT_NONE, qo, torch.float32, want_idx=True, want_y=False)
        us = graph_us(fn)
        print(f'M={M:6d} N={N:5d} K={K:5d} {kind:28s} tiles128={M // 128 * (N // 128):5d} {us:8.1f} us {2.0 * M * N * K / us / 3.944e9 * 100:5.1f} %', flush=True)
