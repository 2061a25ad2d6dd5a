import sys, time, copy
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
import torch
from quantization.quantizers import QMethods
from quantization.base_quantized_model import QuantizedModel
from quantization.autoquant_utils import quantize_model
from quantization.adaround import apply_adaround_to_layer
from quantization.adaround.config import DEFAULT_ADAROUND_CONFIG
dev='cuda'
for (fin,fout,act) in ((768,3072,None),(768,768,None),(3072,768,None),(768,3072,'gelu')):
    class Net(QuantizedModel):
        def __init__(self):
            super().__init__()
            lin=torch.nn.Linear(fin,fout)
            self.fc=quantize_model(torch.nn.Sequential(lin, torch.nn.GELU()) if act else lin, method=QMethods.symmetric_uniform, n_bits=4)
        def forward(self,x): return self.fc(x)
    data=torch.randn(256,128,fin,device=dev)
    for iters in (50, 50, 1000):
        torch.manual_seed(1000)
        net=Net().to(dev)
        layer = net.fc[0] if act else net.fc
        net.set_quant_state(True,False); net.eval()
        with torch.no_grad(): net(data[:8])
        cfg=copy.deepcopy(DEFAULT_ADAROUND_CONFIG); cfg.iters=iters
        net.full_precision(); layer.quantized_weights()
        torch.cuda.synchronize(); t0=time.perf_counter()
        res=apply_adaround_to_layer(net, layer, data, batch_size=8, act_quant=False, adaround_config=cfg)
        torch.cuda.synchronize(); t=time.perf_counter()-t0
        if iters==50: t50=t
    per_iter=(t-t50)/950*1e3
    print(f'Linear({fin},{fout}) act={act}: {per_iter:.3f} ms/iter (marginal over 950 iters), 1000 iters total {t:.3f} s, loss_hard {res.loss_hard_before:.6f}->{res.loss_hard_after:.6f}')
