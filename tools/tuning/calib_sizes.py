"""Estimate + quantize step (running min/max, per-tensor) over tensor sizes around the 256 MiB MALL."""
import sys
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch, os
from quantization.base_quantized_classes import QuantizedActivation
from quantization.quantizers import QMethods
from quantization.range_estimators import RangeEstimators
def ev(fn, n=20, w=5, rounds=3):
    for _ in range(w): fn()
    best = 1e9
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n)
    return best
print('TQ_STATS_NT_MIN_MB =', os.environ.get('TQ_STATS_NT_MIN_MB', '(default 256)'))
for B, S in ((16, 512), (64, 512), (128, 512), (256, 512), (512, 512), (1024, 512)):
    for dt in (torch.bfloat16,):
        x = torch.randn(B, S, 768, device='cuda').to(dt)
        qa = QuantizedActivation(act_method=QMethods.asymmetric_uniform, n_bits_act=8, act_range_method=RangeEstimators.running_minmax).cuda().eval()
        qa.quantized_acts()
        with torch.no_grad():
            qa(x)
            ms = ev(lambda: qa(x))
        mb = x.numel() * x.element_size() / 2**20
        print(f'[{B},{S},768] {str(dt)[6:]} {mb:7.0f} MiB  estimate+quantize {ms*1e3:8.1f} us  {x.numel()*6/ms/1e9:6.2f} TB/s (6 B/elem)')
