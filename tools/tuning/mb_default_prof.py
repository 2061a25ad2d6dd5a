"""Kernel-level picture of the DEFAULT-route MobileBERT W4A4 forward ([8,128], one hipGraph replay = one forward): run under
    rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python tools/tuning/mb_default_prof.py
then  python tools/tuning/layer_timeline.py OUT mobilebert"""
import sys
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from tests.test_mobilebert_e2e import _build, _fixture, _calibrate_and_run
from quantization.graphs import GraphedForward
z = _fixture(); model, hf = _build('cuda'); ids = torch.from_numpy(z['input_ids']).cuda()
_calibrate_and_run(model, ids)
with torch.no_grad():
    g = GraphedForward(model, ids)
    torch.cuda.synchronize()
    for _ in range(20):
        g(ids)
torch.cuda.synchronize()
