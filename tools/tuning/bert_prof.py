import sys
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
import torch
from tests.test_bert_e2e import _build, _fixture, _calibrate_and_run
from tests.harness_bert import QResidualBlock, QSelfAttention
z=_fixture(); model,hf=_build('cuda'); ids=torch.from_numpy(z['input_ids']).cuda()
_calibrate_and_run(model, ids)
mode = sys.argv[1] if len(sys.argv)>1 else 'layered'
QResidualBlock.fuse = mode in ('fused','int8','int8attn')
QSelfAttention.fuse = mode == 'int8attn'
from quantization import options
options.INT8_LINEAR = mode in ('int8','int8attn')
with torch.no_grad():
    for _ in range(20): model(ids)
torch.cuda.synchronize()
