import sys, os, time
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
import numpy as np, torch
from quantization import _hip
from quantization.range_estimators import candidate_params
be=_hip.backend(); dev='cuda'
def timeit(fn, reps=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/reps
# MSE 1D (100 cand) and 2D (100x64x2) on [8,128,768] and bigger
for shape in [(8,128,768),(64,128,768),(256,512,768)]:
    for dt in (torch.float32, torch.bfloat16):
        x=(torch.randn(*shape,device=dev)*3).to(dt); n=x.numel()
        for C,label in ((100,'1D'),(12800,'2D')):
            if C==12800 and n>10_000_000: continue
            pos=np.linspace(0.1,12,C); neg=-pos
            cand=be.candidate_table(candidate_params(neg,pos,8,False),x.device)
            loss=be.zeros_f64((1,C),x.device)
            ms=timeit(lambda: be.mse_candidates(x,1,cand,loss), reps=5)
            flops=n*C*25
            print(f'mse{label} {str(dt)[6:]} {shape}: {ms*1e3:.0f} us  {n/ms/1e3:.1f} Melem/s  {flops/ms/1e9:.1f} TFLOP/s(25 op/elem/cand)', flush=True)
# per-channel MSE on weights [3072,768], 100 candidates
w=torch.randn(3072,768,device=dev)*0.05
cand=be.candidate_table(candidate_params(-np.linspace(0.01,0.3,100),np.linspace(0.01,0.3,100),4,True),dev)
loss=be.zeros_f64((3072,100),dev)
ms=timeit(lambda: be.mse_candidates(w,3072,cand,loss),reps=5)
print(f'mse per-channel [3072,768] C=100: {ms*1e3:.0f} us', flush=True)
# adaround kernels on [3072,768]
alpha=torch.randn(3072,768,device=dev); g=torch.randn(3072,768,device=dev)
m=torch.zeros_like(alpha); v=torch.zeros_like(alpha)
delta=torch.tensor(0.01,device=dev); sg=torch.tensor(True,device=dev)
qargs=(delta,None,sg,4,True,False,1e-8,1,1)
n=w.numel()
ms=timeit(lambda: be.adaround_fwd(w,alpha,qargs,1,True,1.0)); print(f'ada_fwd [3072,768]: {ms*1e3:.1f} us {n*12/ms/1e6:.0f} GB/s')
ms=timeit(lambda: be.adaround_bwd_adam(w,g,alpha,m,v,qargs,1,1.0,0.01,10.0,1e-3,0.9,0.999,1e-8,3)); print(f'ada_bwd_adam: {ms*1e3:.1f} us {n*28/ms/1e6:.0f} GB/s')
ms=timeit(lambda: be.adaround_reg(alpha,1,1.0,10.0,0.01)); print(f'ada_reg: {ms*1e3:.1f} us')
p=torch.randn(8,128,3072,device=dev); t=torch.randn(8,128,3072,device=dev)
ms=timeit(lambda: be.recon_loss(p,t)); print(f'recon [8,128,3072]: {ms*1e3:.1f} us {p.numel()*8/ms/1e6:.0f} GB/s')
# whole calibrating call latency, small tensor
from quantization.quantizers import QMethods
from quantization.range_estimators import RangeEstimators
from quantization.quantization_manager import QuantizationManager
x=torch.randn(8,128,768,device=dev)
for init in ('running_minmax','current_minmax'):
    mgr=QuantizationManager(qmethod=QMethods.asymmetric_uniform, init=RangeEstimators[init], qparams=dict(n_bits=8))
    mgr(x)
    ms=timeit(lambda: mgr(x), reps=50); 
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(200): mgr(x)
    torch.cuda.synchronize(); wall=(time.perf_counter()-t0)/200
    print(f'calibrating call {init} [8,128,768]: {ms*1e3:.1f} us (events) {wall*1e6:.1f} us (wall)')
mgr.fix_ranges()
t0=time.perf_counter()
for _ in range(500): mgr(x)
torch.cuda.synchronize(); print(f'fixed call wall {(time.perf_counter()-t0)/500*1e6:.1f} us')
