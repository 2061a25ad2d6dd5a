"""MobileBERT W4A4 default-route forward as one hipGraph with the four feed-forward blocks of a layer as ONE chained launch
(QMobileLayer.fuse_chain, the default) against one launch per block, over batch sizes."""
import sys, time
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from tests.test_mobilebert_e2e import _build, _fixture, _calibrate_and_run
from harness.mobilebert import QMobileLayer, QFFN
from quantization.graphs import GraphedForward
z = _fixture(); model, hf = _build('cuda')
ids0 = torch.from_numpy(z['input_ids']).cuda()
_calibrate_and_run(model, ids0)
for B in (8, 32, 64, 128, 256):
    ids = ids0.repeat((B + ids0.shape[0] - 1) // ids0.shape[0], 1)[:B].contiguous()
    res = []
    for chain, ffn in ((True, None), (False, None), (False, False)):
        QMobileLayer.fuse_chain = chain
        QFFN.fuse = ffn
        QMobileLayer.fuse_ffn = ffn
        with torch.no_grad():
            g = GraphedForward(model, ids)
            for _ in range(5): g(ids)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): g(ids)
            torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 20 * 1e3)
    print(f'[{B},128]: chained {res[0]:.3f} ms, one launch per block {res[1]:.3f} ms, two integer Linears per block {res[2]:.3f} ms', flush=True)
