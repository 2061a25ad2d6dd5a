"""BERT-large (24 x 1024, 16 heads) W8A8 fixed-range forward, B=8, T=128 / 384: layered vs all fast paths."""
import sys, time
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from transformers import BertConfig, BertForSequenceClassification
from harness.bert import QBertForSequenceClassification, QResidualBlock, QSelfAttention
from quantization import options
from quantization.quantizers import QMethods
from quantization.range_estimators import RangeEstimators
def t(fn, n=10, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
torch.manual_seed(0)
cfg = BertConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096, num_labels=2)
hf = BertForSequenceClassification(cfg).eval()
qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
          weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)
model = QBertForSequenceClassification(hf, **qp).cuda().eval()
hf = hf.cuda()
def graph(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out
for T in (128, 384):
    ids = torch.randint(1000, 30000, (8, T), device='cuda')
    with torch.no_grad():
        model.reset_act_ranges() if T != 128 else None
        model.set_quant_state(True, True)
        model.estimate_ranges()
        model(ids)
        cal = t(lambda: model(ids), n=5)
        model.fix_ranges()
        fp = t(lambda: hf(input_ids=ids).logits)
        g0, o0 = graph(lambda: model(ids)); lay = t(lambda: g0.replay(), n=20)
        QResidualBlock.fuse = QSelfAttention.fuse = True; options.INT8_LINEAR = True
        g1, o1 = graph(lambda: model(ids)); fast = t(lambda: g1.replay(), n=20)
        QResidualBlock.fuse = QSelfAttention.fuse = False; options.INT8_LINEAR = False
    print(f'BERT-large B=8 T={T}: HF fp32 {fp:.2f} ms | calibrating (eager) {cal:.2f} ms | fixed layered graph {lay:.2f} ms | '
          f'integer fast paths graph {fast:.2f} ms | max logit dev {float((o1 - o0).abs().max()):.4f} (span {float(o0.max() - o0.min()):.3f})')
