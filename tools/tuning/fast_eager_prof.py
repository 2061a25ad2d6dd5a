"""Host-side profile of the eager fast forward (all opt-in fixed-range paths on)."""
import cProfile, pstats, io, sys, time
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from tests.test_bert_e2e import _build, _fixture, _calibrate_and_run
from harness.bert import QResidualBlock, QSelfAttention
from quantization import options
z = _fixture(); model, _ = _build('cuda'); ids = torch.from_numpy(z['input_ids']).cuda()
_calibrate_and_run(model, ids)
from harness.bert import QLayer
QResidualBlock.fuse = QSelfAttention.fuse = QLayer.fuse_ffn = None; options.INT8_LINEAR = 'auto'      # the default route
with torch.no_grad():
    for _ in range(5): model(ids)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): model(ids)
    torch.cuda.synchronize(); print('eager fast forward ms', (time.perf_counter() - t0) / 20 * 1e3)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): model(ids)
    torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(40); print(s.getvalue()[:9000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:9000])
