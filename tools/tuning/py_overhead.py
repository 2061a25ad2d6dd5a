"""Where the host time of a small-tensor quantizer call goes (cProfile, launch-bound regime)."""
import cProfile, pstats, sys, time, io
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization.base_quantized_classes import QuantizedActivation
from quantization.quantizers import QMethods
from quantization.range_estimators import RangeEstimators
x = torch.randn(8, 128, 768, device='cuda')
def mk(method=RangeEstimators.running_minmax):
    qa = QuantizedActivation(act_method=QMethods.asymmetric_uniform, n_bits_act=8, act_range_method=method).cuda()
    qa.quantized_acts(); qa.eval(); return qa
def wall(fn, n=3000):
    for _ in range(100): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
with torch.no_grad():
    qa = mk(); qa(x)
    print('calibrating call us', wall(lambda: qa(x)))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3000): qa(x)
    pr.disable(); torch.cuda.synchronize()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18); print(s.getvalue()[:3500])
    qa.fix_ranges()
    print('fixed call us', wall(lambda: qa(x)))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3000): qa(x)
    pr.disable(); torch.cuda.synchronize()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:3000])
