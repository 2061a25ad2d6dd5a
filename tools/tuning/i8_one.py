import sys, os
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
import torch
from quantization import _hip
be=_hip.backend(); dev='cuda'
M,N,K=8192,3072,768
x=torch.randint(-128,127,(M,K),dtype=torch.int8,device=dev); w=torch.randint(-127,127,(N,K),dtype=torch.int8,device=dev)
rs=be.rowsum_i8(w); b=torch.randn(N,device=dev)
xd=torch.tensor(0.02,device=dev); xz=torch.tensor(117.0,device=dev); wd=torch.tensor(0.001,device=dev).reshape(1)
od=torch.tensor(0.05,device=dev); oz=torch.tensor(100.0,device=dev)
mode=sys.argv[1] if len(sys.argv)>1 else 'fused'
q=(od,oz,None,8,False,False,1e-8)
st=be.act_stair(_hip.ACT_GELU,q) if mode.startswith('stair') else None
for _ in range(10):
    if mode=='stair': be.linear_i8(x,w,rs,b,(xd,xz,8,1e-8),wd,1e-8,_hip.ACT_GELU,q,torch.float32, want_idx=True, stair=st)
    elif mode=='stair_idx': be.linear_i8(x,w,rs,b,(xd,xz,8,1e-8),wd,1e-8,_hip.ACT_GELU,q,torch.float32, want_idx=True, want_y=False, stair=st)
    elif mode=='fused_idx': be.linear_i8(x,w,rs,b,(xd,xz,8,1e-8),wd,1e-8,_hip.ACT_GELU,q,torch.float32, want_idx=True, want_y=False)
    elif mode=='fused': be.linear_i8(x,w,rs,b,(xd,xz,8,1e-8),wd,1e-8,_hip.ACT_GELU,(od,oz,None,8,False,False,1e-8),torch.float32, want_idx=True)
    else: be.linear_i8(x,w,rs,b,(xd,xz,8,1e-8),wd,1e-8,_hip.ACT_NONE,None,torch.float32)
torch.cuda.synchronize()
