import sys, time
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
import numpy as np, torch
from tests.test_bert_e2e import _build, _fixture
from utils.utils import pass_data_for_range_estimation
z=_fixture()
model,hf=_build('cuda')
ids=torch.from_numpy(z['input_ids']).cuda()
def t(fn,n=10,w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
with torch.no_grad():
    model.set_quant_state(False, False); model.eval()
    print('fp32 forward ms', t(lambda: model(ids)))
    model.set_quant_state(True, True)
    print('calibrating forward ms', t(lambda: model(ids)))
    model.fix_ranges()
    print('fixed-range forward ms (eager)', t(lambda: model(ids)))
    # hipGraph capture of the fixed-range forward
    g=torch.cuda.CUDAGraph()
    s=torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): model(ids)
    torch.cuda.current_stream().wait_stream(s)
    try:
        with torch.cuda.graph(g):
            out=model(ids)
        print('fixed-range forward ms (hipGraph replay)', t(lambda: g.replay()))
        ref=model(ids)
        print('graph == eager', torch.equal(out, ref))
    except Exception as e:
        print('graph capture failed:', repr(e)[:300])
from tests.harness_bert import QResidualBlock
QResidualBlock.fuse=True
with torch.no_grad():
    print('fixed-range forward ms (eager, fused tails)', t(lambda: model(ids)))
    g2=torch.cuda.CUDAGraph()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): model(ids)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g2):
        out2=model(ids)
    print('fixed-range forward ms (hipGraph, fused tails)', t(lambda: g2.replay(), n=30))
from quantization import options
options.INT8_LINEAR=True
with torch.no_grad():
    print('fixed-range forward ms (eager, int8 linears + fused tails)', t(lambda: model(ids)))
    g3=torch.cuda.CUDAGraph()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): model(ids)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g3):
        out3=model(ids)
    print('fixed-range forward ms (hipGraph, int8 linears + fused tails)', t(lambda: g3.replay(), n=30))
    print('max |logit diff| vs fp32-simulated GEMMs', float((out3-out).abs().max()))
from tests.harness_bert import QSelfAttention
QSelfAttention.fuse=True
with torch.no_grad():
    print('fixed-range forward ms (eager, int8 + fused tails + fused attention probs)', t(lambda: model(ids)))
    g4=torch.cuda.CUDAGraph()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): model(ids)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g4):
        out4=model(ids)
    print('fixed-range forward ms (hipGraph, int8 + fused tails + fused attention probs)', t(lambda: g4.replay(), n=30))
    print('max |logit diff| vs fp32-simulated GEMMs', float((out4-out).abs().max()))
print('(the last configuration above already routes the attention core through tq_attention_i8_fwd when the tags are present)')
# ---- calibration as a hipGraph (options.INPLACE_CALIBRATION_STATE) ----
QResidualBlock.fuse = QSelfAttention.fuse = False
options.INT8_LINEAR = False
options.INPLACE_CALIBRATION_STATE = True
model2, _ = _build('cuda')
with torch.no_grad():
    model2.set_quant_state(True, True)
    model2(ids)
    print('calibrating forward ms (eager, in-place state)', t(lambda: model2(ids)))
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): model2(ids)
    torch.cuda.current_stream().wait_stream(s)
    gc_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gc_):
        oc = model2(ids)
    print('calibrating forward ms (hipGraph replay)', t(lambda: gc_.replay(), n=30))
