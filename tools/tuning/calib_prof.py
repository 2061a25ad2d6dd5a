"""cProfile of the launch-bound CALIBRATING quantizer call (running min/max, [8,128,768], per-tensor): where the ~19 us of host time go."""
import cProfile, pstats, sys, time
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization.base_quantized_classes import QuantizedActivation
from quantization.quantizers import QMethods
from quantization.range_estimators import RangeEstimators
from quantization import options
x = torch.randn(8, 128, 768, device='cuda')
for inplace in (False, True):
    options.INPLACE_CALIBRATION_STATE = inplace
    qa = QuantizedActivation(act_method=QMethods.asymmetric_uniform, n_bits_act=8, act_range_method=RangeEstimators.running_minmax).cuda()
    qa.quantized_acts(); qa.eval()
    with torch.no_grad():
        for _ in range(300): qa(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5000): qa(x)
        torch.cuda.synchronize(); print('inplace', inplace, 'us per call', round((time.perf_counter() - t0) / 5000 * 1e6, 2))
        pr = cProfile.Profile(); pr.enable()
        for _ in range(3000): qa(x)
        torch.cuda.synchronize(); pr.disable()
    st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(14)
