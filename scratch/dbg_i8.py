import sys
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
import torch
from quantization import options
from tests.test_bert_e2e import _build, _fixture, _calibrate_and_run
z=_fixture(); model,_=_build('cuda'); ids=torch.from_numpy(z['input_ids']).cuda()
layered=_calibrate_and_run(model, ids)
options.INT8_LINEAR=True
q=model.layers[0].attention_self.query
with torch.no_grad():
    h=model.embeddings(ids)
    print('tag', getattr(h,'_tq_quantizer',None))
    print('int8 fwd', type(q._int8_forward(h)))
    src=h._tq_quantizer if hasattr(h,'_tq_quantizer') else None
    w=q.weight_quantizer
    print(q.training, q._quant_w, type(q.activation_function), h.dtype, w.state, w.quantizer.symmetric, w.quantizer.n_bits, w.quantizer._delta.numel(), src.symmetric if src else None)
