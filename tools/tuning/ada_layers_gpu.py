import sys
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
import torch, copy

from quantization import _hip

from quantization.base_quantized_model import QuantizedModel
from quantization.autoquant_utils import quantize_model
from quantization.adaround import apply_adaround_to_layer
from quantization.adaround.config import DEFAULT_ADAROUND_CONFIG
from quantization.quantizers import QMethods
class Net(QuantizedModel):
    def __init__(self):
        super().__init__()
        self.emb = quantize_model(torch.nn.Embedding(100, 32), method=QMethods.symmetric_uniform, n_bits=4)
        self.ln = quantize_model(torch.nn.LayerNorm(32), method=QMethods.symmetric_uniform, n_bits=4)
        self.fc = quantize_model(torch.nn.Linear(32, 16), method=QMethods.symmetric_uniform, n_bits=4)
    def forward(self, ids):
        return self.fc(self.ln(self.emb(ids)))
torch.manual_seed(0)
net = Net().eval().cuda()
data = torch.randint(0, 100, (16, 8)).cuda()
net.set_quant_state(True, False)
with torch.no_grad(): net(data[:4])
for name in sys.argv[1:]:
    cfg = copy.deepcopy(DEFAULT_ADAROUND_CONFIG); cfg.iters = 10
    net.full_precision(); getattr(net, name).quantized_weights()
    try:
        r = apply_adaround_to_layer(net, getattr(net, name), data, batch_size=4, act_quant=False, adaround_config=cfg)
        print(name, 'ok', r.loss_hard_before, r.loss_hard_after)
    except Exception as e:
        import traceback; traceback.print_exc(); print(name, 'FAILED', repr(e)[:200])
