import sys
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
import torch
from quantization import _hip
be=_hip.backend()
n=4096
miss=[]
for k in range(0,n,1):
    x=torch.ones(n, device='cuda'); x[k]=-100
    mn,mx=be.minmax(x,1,1)
    if float(mn)!=-100: miss.append(k)
print('missed', len(miss), miss[:64])
import collections
print(collections.Counter([(k//4)//64 for k in miss]).most_common(20))
print(collections.Counter([k%4 for k in miss]))
