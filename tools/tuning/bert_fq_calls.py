"""Which quantizer modules still launch a stand-alone fake-quant kernel in the BERT-base fast forward."""
import sys, collections
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
import torch
from tests.test_bert_e2e import _build, _fixture, _calibrate_and_run
from tests.harness_bert import QResidualBlock, QSelfAttention
from quantization import options
from quantization.quantization_manager import QuantizationManager
z=_fixture(); model,hf=_build('cuda'); ids=torch.from_numpy(z['input_ids']).cuda()
_calibrate_and_run(model, ids)
QResidualBlock.fuse = True; QSelfAttention.fuse = True; options.INT8_LINEAR = True
names = {m: n for n, m in model.named_modules()}
cnt = collections.Counter()
orig = QuantizationManager.forward
def spy(self, x):
    n = names.get(self, '?')
    cnt['.'.join(n.split('.')[2:]) if n.startswith('layers.') else n] += 1
    return orig(self, x)
with torch.no_grad():
    model(ids)
    QuantizationManager.forward = spy
    model(ids)
QuantizationManager.forward = orig
for k, v in cnt.most_common(): print(v, k)
