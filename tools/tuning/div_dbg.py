import sys, numpy as np, torch
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
from tests.test_mobilebert_e2e import _build, _fixture, _calibrate_and_run, _IntegerMode
from harness.divergence import encoder_flip_rates
from quantization.graphs import GraphedForward
from quantization import options
z=_fixture(); model,hf=_build('cuda'); ids=torch.from_numpy(z['input_ids'])
_calibrate_and_run(model, ids)
ids=ids.cuda()
def show(tag):
    rows,first=encoder_flip_rates(model, ids, _IntegerMode())
    print(tag,'same',[round(r['same_input']['flip_rate'],5) for r in rows][:8],'free',[round(r['free_running']['flip_rate'],4) for r in rows][:5])
show('fresh')
show('again')
with torch.no_grad():
    g=GraphedForward(model, ids); g(ids)
show('after layered graph')
with _IntegerMode():
    with torch.no_grad():
        g2=GraphedForward(model, ids); g2(ids)
show('after integer graph')
