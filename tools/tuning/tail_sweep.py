"""Fused residual tails: TQ_TAIL_ITERS sweep at HBM-bound sizes + whole-BERT fused-vs-layered agreement."""
import os, sys
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization import _hip
be = _hip.backend()
def ev(fn, n=20, w=5, rounds=3):
    for _ in range(w): fn()
    best = 1e9
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n)
    return best
q = lambda d_, z_: (torch.tensor(d_, device='cuda'), torch.tensor(z_, device='cuda'), None, 8, False, False, 1e-8)
for rows, d in ((131072, 768), (131072, 512), (1024, 768)):
    for dt in (torch.bfloat16, torch.float32):
        a = torch.randn(rows, d, device='cuda').to(dt); r = torch.randn(rows, d, device='cuda').to(dt)
        w = torch.randn(d, device='cuda'); b = torch.randn(d, device='cuda')
        q1, q2, q3 = q(0.05, 120.0), q(0.06, 128.0), q(0.03, 128.0)
        for name, eps in (('LayerNorm', 1e-12), ('NoNorm', None)):
            res = []
            for it in (1, 2, 4, 8, 16):
                os.environ['TQ_TAIL_ITERS'] = str(it)
                ms = ev(lambda: be.residual_layernorm_quant(a, r, q1, q2, w, b, eps, q3))
                res.append(f'it={it}: {ms*1e3:7.1f} us {3*a.numel()*a.element_size()/ms/1e9:5.2f} TB/s')
            print(f'[{rows},{d}] {str(dt)[6:]:9s} {name:9s} ' + ' | '.join(res), flush=True)
os.environ.pop('TQ_TAIL_ITERS')
from tests.test_bert_e2e import _build, _fixture, _calibrate_and_run
from tests.harness_bert import QResidualBlock
z = _fixture(); model, _ = _build('cuda'); ids = torch.from_numpy(z['input_ids']).cuda()
layered = _calibrate_and_run(model, ids)
QResidualBlock.fuse = True
with torch.no_grad(): fused = model(ids)
span = float(layered.max() - layered.min())
print('bert fused vs layered: max dev / span', float((fused - layered).abs().max()) / span, 'identical frac', float(((fused - layered).abs() == 0).float().mean()))
print(layered[:3], fused[:3])
