import sys, os
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
import torch
from quantization import _hip
be=_hip.backend(); dev='cuda'
def graph_time(fn, n=20):
    s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s)
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/5/n*1e3
for (M,N,K) in [(1024,768,768),(1024,3072,768),(1024,768,3072),(8192,768,768),(8192,3072,768)]:
    x=torch.randint(-128,127,(M,K),dtype=torch.int8,device=dev); w=torch.randint(-127,127,(N,K),dtype=torch.int8,device=dev)
    rs=be.rowsum_i8(w); b=torch.randn(N,device=dev)
    xd=torch.tensor(0.02,device=dev); xz=torch.tensor(117.0,device=dev); wd=torch.tensor(0.001,device=dev).reshape(1)
    od=torch.tensor(0.05,device=dev); oz=torch.tensor(100.0,device=dev)
    t_i8=graph_time(lambda: be.linear_i8(x,w,rs,b,(xd,xz,8,1e-8),wd,1e-8,_hip.ACT_GELU,(od,oz,None,8,False,False,1e-8),torch.float32))
    t_plain=graph_time(lambda: be.linear_i8(x,w,rs,b,(xd,xz,8,1e-8),wd,1e-8,_hip.ACT_NONE,None,torch.float32))
    t_q=graph_time(lambda: be.linear_i8(x,w,rs,b,(xd,xz,8,1e-8),wd,1e-8,_hip.ACT_NONE,(od,oz,None,8,False,False,1e-8),torch.float32))
    print(f'   plain (no act, no quant) {t_plain:.1f} us | quant only {t_q:.1f} us')
    xf=torch.randn(M,K,device=dev); wf=torch.randn(N,K,device=dev)
    t_f32=graph_time(lambda: torch.nn.functional.linear(xf,wf,b))
    xb=xf.bfloat16(); wb=wf.bfloat16(); bb=b.bfloat16()
    t_bf=graph_time(lambda: torch.nn.functional.linear(xb,wb,bb))
    print(f'M={M} N={N} K={K}: i8 fused {t_i8:.1f} us ({2*M*N*K/t_i8/1e6:.0f} TOPS) | torch fp32 linear {t_f32:.1f} us | torch bf16 linear {t_bf:.1f} us', flush=True)
