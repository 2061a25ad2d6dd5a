import sys
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization import _hip
be = _hip.backend()
for dt in (torch.float32, torch.bfloat16):
    for shape in ((8, 128, 768), (8, 128, 3072), (8, 12, 128, 128)):
        x = torch.randn(*shape, device='cuda').to(dt)
        prev = None
        for _ in range(60):
            r = be.calibrate_minmax(x, 1, 1, 2, None if prev is None else prev[0], None if prev is None else prev[1], 0.9, 0, None, 8, False, 1e-8, False)
            prev = r
        for _ in range(60):
            be.fake_quant(x, r[2], r[3], None, 8, False, False, 1e-8, 1, 1)
            be.minmax(x, 1, 1)
torch.cuda.synchronize()
