"""Which MobileBERT Linears do NOT take the integer path in the fast forward, and why (first failing condition)."""
import sys, collections
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization import options, provenance
from quantization.autoquant_utils import QuantLinear, INT8_STATS
from tests.test_mobilebert_e2e import _build as _build_mb, _fixture as _fixture_mb
from harness.mobilebert import QResidualNoNorm, QMobileSelfAttention
zm = _fixture_mb()
mb, _ = _build_mb('cuda')
ids = torch.from_numpy(zm['input_ids']).cuda()
names = {m: n for n, m in mb.named_modules()}
fall = collections.Counter()
orig = QuantLinear._int8_forward
def spy(self, x, with_output_quantizer=True):
    y = orig(self, x, with_output_quantizer)
    if y is None:
        src = provenance.quantizer_of(x)
        why = 'no provenance' if src is None else ('src symmetric' if src.symmetric else ('src vector' if src._delta.numel() != 1 else 'other'))
        n = names[self]
        fall[(n.split('.')[-2] + '.' + n.split('.')[-1] if 'layers' in n else n, why, tuple(x.shape[-1:]), self.out_features)] += 1
    return y
QuantLinear._int8_forward = spy
with torch.no_grad():
    mb.set_quant_state(True, True); mb(ids); mb.fix_ranges()
    QResidualNoNorm.fuse = True; QMobileSelfAttention.fuse = True; options.INT8_LINEAR = True
    mb(ids); fall.clear(); k0 = INT8_STATS['kernel_calls']
    mb(ids)
print('int8 kernel calls per forward', INT8_STATS['kernel_calls'] - k0)
for k, v in fall.most_common():
    print(v, k)
