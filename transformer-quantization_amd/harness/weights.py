"""Deterministic model parameters that do not depend on the torch / transformers build.

The whole-model parity fixtures (tests/golden/*bert*.npz) hold the reference's RESULTS, not the 100-440 MB of weights
they were computed on; both sides -- the fixture generators that import the reference and the harness models the tests
run -- regenerate the weights.  Seeding torch's generator and relying on the HuggingFace initialisers made that depend on
the initialisation order of the installed transformers release and on torch's sampling kernels: a different build on the
GPU box meant different weights and a test that could only skip.  Here every parameter comes from
numpy's legacy Mersenne-Twister stream (`np.random.RandomState(...).standard_normal`, bit-stable across numpy versions
and platforms), seeded per parameter from (seed, qualified parameter name), so neither the order in which modules are
created nor parameters added by a newer release change any value.

Distributions follow the HuggingFace BERT initialiser (initializer_range 0.02): matrices N(0, 0.02^2), biases 0,
LayerNorm weight 1 / bias 0.  MobileBERT's NoNorm gets weight 1 + 0.1 N(0,1), bias 0.5 N(0,1): the reference's
QuantNoNorm quantizes weight AND bias with ONE quantizer whose range ends up being the bias range (quirk q9,
reference models/quantized_mobilebert.py:58-72) -- with HF's all-zero bias every NoNorm weight would quantize to ~0 and
the network would output zeros; a trained checkpoint has non-trivial affine parameters.

This file imports numpy and torch only: tests/golden/make_golden_*.py load it by path next to the imported reference.
"""
import zlib

import numpy as np
import torch


def fill_from_numpy_stream(module, seed):
    """Overwrite every parameter of `module` in place; returns the module."""
    with torch.no_grad():
        for mname, m in module.named_modules():
            kind = type(m).__name__
            for pname, p in m.named_parameters(recurse=False):
                name = f'{mname}.{pname}' if mname else pname
                rs = np.random.RandomState(zlib.crc32(f'{int(seed)}:{name}'.encode()) & 0xFFFFFFFF)
                if kind == 'NoNorm':
                    v = (1.0 + 0.1 * rs.standard_normal(p.shape)) if pname == 'weight' else 0.5 * rs.standard_normal(p.shape)
                elif p.dim() >= 2:
                    v = 0.02 * rs.standard_normal(p.shape)
                elif pname == 'weight':
                    v = np.ones(p.shape)
                else:
                    v = np.zeros(p.shape)
                p.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)).to(p.dtype))
    return module


def weight_check_sum(module):
    """float64 sum of |w| over all parameters, in name order, accumulated with numpy's pairwise sum on the CPU
    (thread-count independent): the value the fixtures store to prove both sides ran on the same weights."""
    total = 0.0
    for _, p in sorted(module.named_parameters()):
        total += float(np.abs(p.detach().cpu().double().numpy()).sum())
    return total
