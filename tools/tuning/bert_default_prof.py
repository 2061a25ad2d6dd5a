"""Kernel-level picture of the DEFAULT-route BERT-base forward ([8,128], one hipGraph replay = one forward): run under
    rocprofv3 --kernel-trace --stats -d OUT -o t -- python tools/tuning/bert_default_prof.py
and read OUT/*kernel_stats.csv (20 replays)."""
import sys
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from tests.test_bert_e2e import _build, _fixture, _calibrate_and_run
from quantization.graphs import GraphedForward
z = _fixture(); model, hf = _build('cuda'); ids = torch.from_numpy(z['input_ids']).cuda()
_calibrate_and_run(model, ids)
with torch.no_grad():
    g = GraphedForward(model, ids)
    torch.cuda.synchronize()
    for _ in range(20):
        g(ids)
torch.cuda.synchronize()
