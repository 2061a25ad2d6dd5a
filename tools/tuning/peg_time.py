"""PEG-6 / per-embedding BERT-base forward timings (calibrating and fixed-range)."""
import sys, time
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from tests.test_bert_e2e import _build, _fixture
from harness.bert import apply_activation_granularity, estimate_permutation_ranges
z = _fixture()
ids = torch.from_numpy(z['input_ids']).cuda()
def t(fn, n=10, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name, kw in (('per-tensor', {}), ('per-embd', dict(per_embd=True)), ('PEG-6', dict(per_groups=6)),
                 ('PEG-6 permuted', dict(per_groups=6, permute=True))):
    model, _ = _build('cuda')
    apply_activation_granularity(model, **kw)
    with torch.no_grad():
        if kw.get('permute'):
            estimate_permutation_ranges(model, [(ids,)])
        model.set_quant_state(True, True)
        model(ids)
        cal = t(lambda: model(ids))
        model.fix_ranges()
        fix = t(lambda: model(ids))
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): model(ids)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            model(ids)
        gr = t(lambda: g.replay(), n=30)
    print(f'{name:16s} calibrating {cal:6.2f} ms   fixed eager {fix:6.2f} ms   fixed hipGraph {gr:6.2f} ms')
