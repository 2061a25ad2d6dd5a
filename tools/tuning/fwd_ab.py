"""Default-route hipGraph forwards of both harness models at [8,128] with whatever library TQ_LIB_PATH points at (A/B of two
builds on one box): prints `bert_ms mobilebert_ms` as the minimum of 5 x 50 replays."""
import sys, time
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization.graphs import GraphedForward
res = []
for mod in ('tests.test_bert_e2e', 'tests.test_mobilebert_e2e'):
    m = __import__(mod, fromlist=['x'])
    z = m._fixture(); model, hf = m._build('cuda'); ids = torch.from_numpy(z['input_ids']).cuda()
    m._calibrate_and_run(model, ids)
    with torch.no_grad():
        g = GraphedForward(model, ids)
        for _ in range(20): g(ids)
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): g(ids)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 50 * 1e3)
    res.append(best)
print('forward_ms bert %.4f mobilebert %.4f' % tuple(res))
