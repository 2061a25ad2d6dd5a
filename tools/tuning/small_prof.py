import sys
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
import torch
from quantization import _hip
be=_hip.backend()
delta=torch.tensor(0.03,device='cuda'); zf=torch.tensor(128.0,device='cuda')
for shape in [(8,128,768),(8,12,128,128),(8,128,3072),(8,768),(64,128,768)]:
    for dt in (torch.float32, torch.bfloat16):
        x=torch.randn(*shape,device='cuda').to(dt)
        for _ in range(30): be.fake_quant(x,delta,zf,None,8,False,False,1e-8,1,1)
        for _ in range(30): be.minmax(x,1,1)
torch.cuda.synchronize()
