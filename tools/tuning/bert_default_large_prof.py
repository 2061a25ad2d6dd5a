"""Default-route BERT-base (or, with a second argument `mobilebert`, MobileBERT W4A4) forward at serving batch sizes ([B,128], B
from argv, default 64) as one hipGraph: wall time per forward
and -- under `rocprofv3 --kernel-trace --stats -d OUT -o t -- python tools/tuning/bert_default_large_prof.py 64` -- the
per-kernel picture (OUT/*kernel_stats.csv, 20 replays)."""
import sys, time
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
import importlib
MODEL = sys.argv[2] if len(sys.argv) > 2 else 'bert'
_m = importlib.import_module('tests.test_mobilebert_e2e' if MODEL == 'mobilebert' else 'tests.test_bert_e2e')
_build, _fixture, _calibrate_and_run = _m._build, _m._fixture, _m._calibrate_and_run
from quantization.graphs import GraphedForward
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
z = _fixture(); model, hf = _build('cuda')
ids0 = torch.from_numpy(z['input_ids']).cuda()
ids = ids0.repeat((B + ids0.shape[0] - 1) // ids0.shape[0], 1)[:B].contiguous()
_calibrate_and_run(model, ids)
with torch.no_grad():
    g = GraphedForward(model, ids)
    for _ in range(5): g(ids)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g(ids)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
print(f'{MODEL} default-route forward [{B},128]: {ms:.3f} ms = {B * 128 / ms * 1e3:.0f} tokens/s', flush=True)
