import sys
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
import torch
from tests.test_bert_e2e import _build, _fixture, _calibrate_and_run
from tests.harness_bert import QResidualBlock
z=_fixture(); model,hf=_build('cuda'); ids=torch.from_numpy(z['input_ids']).cuda()
_calibrate_and_run(model, ids)
QResidualBlock.fuse = len(sys.argv)>1 and sys.argv[1] in ('fused','int8')
from quantization import options
options.INT8_LINEAR = len(sys.argv)>1 and sys.argv[1]=='int8'
with torch.no_grad():
    for _ in range(20): model(ids)
torch.cuda.synchronize()
