import sys, ctypes as C
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
import torch
from quantization import _hip
be=_hip.backend(); lib=be.lib
n=1024*512*768; nb=n*2
big=torch.empty(2*nb+(64<<20), dtype=torch.uint8, device='cuda')
base=big.data_ptr()
base=(base+ (1<<21)-1)//(1<<21)*(1<<21)   # 2MB align
xv=torch.randn(n,device='cuda').to(torch.bfloat16)
import ctypes
delta=torch.tensor(0.03,device='cuda'); zf=torch.tensor(128.0,device='cuda')
q=_hip.tq_quantizer(delta.data_ptr(), zf.data_ptr(), None, 8,0,0,1e-8,1,1)
st=torch.cuda.current_stream().cuda_stream
def run(xp, yp, reps=10):
    for _ in range(2): lib.tq_fake_quant_fwd(xp, yp, None, 0, n, 1, C.byref(q), st)
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): lib.tq_fake_quant_fwd(xp, yp, None, 0, n, 1, C.byref(q), st)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/reps
# copy x into big at base
hipMemcpy=torch.cuda
xb=torch.frombuffer  # unused
view=big[(base-big.data_ptr()):(base-big.data_ptr())+nb].view(torch.bfloat16)
view.copy_(xv)
print('x ptr %x'%base)
for off in [0, 256, 4096, 65536, 1<<20, 2<<20, (2<<20)+4096, 3<<20, 8<<20, (8<<20)+(1<<19), 16<<20, 32<<20, (32<<20)+12345*256]:
    yp=base+nb+off
    ms=run(base, yp)
    print(f'y-x = nb + {off:>10d}: {ms*1e3:8.1f} us  {n*4/ms/1e6:6.0f} GB/s')
# torch-allocated separate tensors
x2=xv.clone(); y2=torch.empty_like(x2)
print('torch alloc: x %x y %x diff %d'%(x2.data_ptr(), y2.data_ptr(), y2.data_ptr()-x2.data_ptr()))
print('torch alloc pair: %.1f us'%(run(x2.data_ptr(), y2.data_ptr())*1e3))
