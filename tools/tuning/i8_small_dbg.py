"""Where the time of the integer Linear goes at the LATENCY-bound shapes of a BERT-base forward at batch 8 (attention
output 1024 x 768 x 768, second feed-forward Linear 1024 x 768 x 3072; plain fp32 output, no quantizer).  Kernel durations
come from rocprofv3, one run per TQ_I8_DBG mode (bit 1 = no epilogue, 2 = no operand loads, 4 = no MFMA):
    TQ_LIB_PATH=.../libtq_hip_dbg.so TQ_I8_DBG=<mode> rocprofv3 --kernel-trace --stats ... -- python tools/tuning/i8_small_dbg.py
(breakdown build: hipcc ... -DTQ_I8_DBG_BUILD csrc/tq_linear_i8.hip)."""
import sys
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import torch
from quantization import _hip
be = _hip.backend(); dev = 'cuda'
for M, N, K in ((1024, 768, 768), (1024, 768, 3072)):
    x = torch.randint(-128, 127, (M, K), dtype=torch.int8, device=dev); w = torch.randint(-127, 127, (N, K), dtype=torch.int8, device=dev)
    rs = be.rowsum_i8(w); b = torch.randn(N, device=dev)
    xd = torch.tensor(0.02, device=dev); xz = torch.tensor(117.0, device=dev); wd = torch.tensor(0.001, device=dev).reshape(1)
    for _ in range(40):
        y = be.linear_i8(x, w, rs, b, (xd, xz, 8, 1e-8), wd, 1e-8, _hip.ACT_NONE, None, torch.float32)
    torch.cuda.synchronize()
