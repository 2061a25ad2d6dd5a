"""Ordered MSE candidate search at the config shape [8,128,768] x 100 candidates: unit size (TQ_ORD_KTOP) x candidate-tile
width (TQ_ORD_NC), HIP-event time of the whole call (unit + fold + row kernels) as a hipGraph of 20 calls (device time)."""
import os, sys
sys.path.insert(0, '/root/repo/transformer-quantization_amd'); sys.path.insert(0, '/root/repo')
import numpy as np, torch
from quantization import _hip
be = _hip.backend()
for shape, C in (((8, 128, 768), 100), ((8, 128, 3072), 100), ((64, 128, 768), 100)):
    x = torch.randn(*shape, device='cuda')
    tab = torch.tensor(np.stack([np.linspace(0.01, 0.2, C), np.full(C, 100.0), np.zeros(C), np.full(C, 255.0)], 1).astype(np.float32)).cuda()
    for kt, nc in ((0, 0), (2, 2), (1, 2), (1, 4), (1, 8)):
        os.environ['TQ_ORD_KTOP'] = str(kt)
        if nc: os.environ['TQ_ORD_NC'] = str(nc)
        else: os.environ.pop('TQ_ORD_NC', None)
        loss = be.zeros_f64((1, C), 'cuda')
        for _ in range(3):
            be.mse_candidates_ordered(x, tab, loss)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                be.mse_candidates_ordered(x, tab, loss)
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): g.replay()
        b.record(); torch.cuda.synchronize()
        print(f'{shape} C={C} ktop={kt} nc={nc or "auto"}: {a.elapsed_time(b) / 100 * 1e3:7.2f} us per search')
