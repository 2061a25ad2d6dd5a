import sys, os
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
import torch
from quantization import _hip
be=_hip.backend()
dev='cuda'
def timeit(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/reps
shapes=[(8,128,768),(64,128,768),(256,512,768),(1024,512,768)]
if len(sys.argv)>1: shapes=[tuple(int(v) for v in sys.argv[1].split(','))]
for dt,es in ((torch.bfloat16,2),(torch.float32,4)):
    for shape in shapes:
        x=torch.randn(*shape, device=dev, dtype=torch.float32).to(dt)
        n=x.numel(); d=shape[-1]
        delta=torch.tensor(0.03,device=dev); zf=torch.tensor(128.0,device=dev)
        dv=torch.full((d,),0.03,device=dev); zv=torch.full((d,),128.0,device=dev)
        r={}
        r['fq_tensor']=(timeit(lambda: be.fake_quant(x,delta,zf,None,8,False,False,1e-8,1,1)), 2*es)
        r['fq_axis']=(timeit(lambda: be.fake_quant(x,dv,zv,None,8,False,False,1e-8,d,1)), 2*es)
        r['fq_idx8']=(timeit(lambda: be.fake_quant(x,delta,zf,None,8,False,False,1e-8,1,1,want_y=False,idx_dtype=torch.uint8)), es+1)
        gy=torch.randn_like(x)
        r['fq_bwd']=(timeit(lambda: be.fake_quant_bwd(x,gy,delta,zf,None,8,False,False,1e-8,1,1)), 3*es)
        r['mm_tensor']=(timeit(lambda: be.minmax(x,1,1)), es)
        r['mm_axis']=(timeit(lambda: be.minmax(x,d,1)), es)
        print(str(dt).split('.')[-1], shape, '  '.join(f'{k}: {ms*1e3:.1f}us {n*b/ms/1e6:.0f}GB/s' for k,(ms,b) in r.items()), flush=True)
