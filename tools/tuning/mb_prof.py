"""Kernel mix of the MobileBERT W4A4 fast forward (fused NoNorm tails + integer Linears), eager launches under rocprofv3:
    rocprofv3 --kernel-trace --stats -d /tmp/mbp -o mb --output-format csv -- python tools/tuning/mb_prof.py"""
import sys
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization import options
from tests.test_mobilebert_e2e import _build as _build_mb, _fixture as _fixture_mb
from harness.mobilebert import QResidualNoNorm
zm = _fixture_mb()
mb, _ = _build_mb('cuda')
ids = torch.from_numpy(zm['input_ids']).cuda()
mode = sys.argv[1] if len(sys.argv) > 1 else 'fast'
with torch.no_grad():
    mb.set_quant_state(True, True)
    mb(ids)
    mb.fix_ranges()
    if mode == 'fast':
        from harness.mobilebert import QBottleneckLayer, QFFN, QMobileSelfAttention
        QResidualNoNorm.fuse = True
        QMobileSelfAttention.fuse = True
        QBottleneckLayer.fuse = True
        QFFN.fuse = True
        options.INT8_LINEAR = True
    for _ in range(3):
        mb(ids)
    torch.cuda.synchronize()
    print('MARK')
    for _ in range(40):
        mb(ids)
    torch.cuda.synchronize()
