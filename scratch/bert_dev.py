import sys
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
import numpy as np, torch
from tests.test_bert_e2e import _build, _fixture, _calibrate_and_run
from tests.harness_bert import quantizer_census
z=_fixture()
model,hf=_build('cuda')
ids=torch.from_numpy(z['input_ids'])
logits=_calibrate_and_run(model, ids)
act,wts=quantizer_census(model)
amin=np.array([float(m.range_estimator.current_xmin) for _,m in act],np.float32)
amax=np.array([float(m.range_estimator.current_xmax) for _,m in act],np.float32)
span=z['act_max']-z['act_min']
rel=np.maximum(np.abs(amin-z['act_min']),np.abs(amax-z['act_max']))/span
print('rel dev: max %.4f median %.6f p90 %.5f'%(rel.max(), np.median(rel), np.percentile(rel,90)))
worst=np.argsort(-rel)[:5]
for i in worst: print(i, act[i][0], rel[i], amin[i], z['act_min'][i], amax[i], z['act_max'][i])
print('first exact sites:', int(np.argmax(rel>0)))
print('logits diff max', np.abs(logits.cpu().numpy()-z['logits']).max(), 'step', (z['act_max'][-1]-z['act_min'][-1])/255)
