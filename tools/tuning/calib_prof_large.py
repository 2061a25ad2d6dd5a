"""Kernel mix of a BERT-base calibrating forward at [128,128] tokens (product defaults): run under
    rocprofv3 --kernel-trace --stats -d OUT -o t -- python tools/tuning/calib_prof_large.py"""
import sys
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from harness.bert import build_bert_base
from quantization.quantizers import QMethods
from quantization.range_estimators import RangeEstimators
qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
          weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)
model, _ = build_bert_base(seed=1000, **qp)
model = model.cuda().eval()
ids = torch.randint(1000, 30000, (128, 128), device='cuda')
with torch.no_grad():
    model.set_quant_state(True, True)
    model.estimate_ranges()
    for _ in range(6):
        model(ids)
torch.cuda.synchronize()
