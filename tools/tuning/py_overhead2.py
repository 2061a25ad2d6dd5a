"""Host time of a launch-bound quantizer call with the fixed-range fast path on / off, and of the calibrating call."""
import sys, time
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization import quantization_manager as qm
from quantization.base_quantized_classes import QuantizedActivation
from quantization.quantizers import QMethods
from quantization.range_estimators import RangeEstimators
x = torch.randn(8, 128, 768, device='cuda')
def mk():
    qa = QuantizedActivation(act_method=QMethods.asymmetric_uniform, n_bits_act=8, act_range_method=RangeEstimators.running_minmax).cuda()
    qa.quantized_acts(); qa.eval(); return qa
def wall(fn, n=5000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
with torch.no_grad():
    qa = mk(); qa(x)
    print('calibrating call us', round(wall(lambda: qa(x)), 2))
    qa.fix_ranges()
    for rep in range(2):
        qm.FAST_FIXED_FORWARD = True
        print('fixed call, fast path us', round(wall(lambda: qa(x)), 2))
        qm.FAST_FIXED_FORWARD = False
        print('fixed call, generic us  ', round(wall(lambda: qa(x)), 2))
    qm.FAST_FIXED_FORWARD = True
    print('torch baseline: x * 2.0 us', round(wall(lambda: x * 2.0), 2))
