"""Helpers that decode the golden fixtures (shared by the CPU oracle tests and the GPU parity tests)."""
import numpy as np
import torch

LAYOUT_ARGS = {
    'per_tensor': dict(axis=None, n_groups=None, per_channel=False),
    'per_embd': dict(axis=2, n_groups=None, per_channel=False),
    'peg6': dict(axis=2, n_groups=6, per_channel=False),
    'peg6_perm': dict(axis=2, n_groups=6, per_channel=False),
    'peg4': dict(axis=2, n_groups=4, per_channel=False),
    'per_channel': dict(axis=None, n_groups=None, per_channel=True),
}


def t(a):
    a = np.asarray(a)
    return torch.from_numpy(a.copy() if a.ndim else np.array(a))   # keeps 0-D shapes


def bf16_from_bits(bits):
    return torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16)


def fq_case(z, m):
    """-> dict with torch tensors for one fake_quant.npz case."""
    k = m['k']
    c = dict(m)
    if m['io'] == 'bf16':
        c['x'] = bf16_from_bits(z[f'c{k}_x'])
        c['y_bf16'] = bf16_from_bits(z[f'c{k}_y_bf16'])
    else:
        c['x'] = t(z[f'c{k}_x'])
    for name in ('xmin', 'xmax', 'delta', 'idx', 'y'):
        c[name] = t(z[f'c{k}_{name}'])
    c['zero_float'] = t(z[f'c{k}_zero_float']) if f'c{k}_zero_float' in z.files else None
    c['ranges'] = t(z[f'c{k}_ranges']) if f'c{k}_ranges' in z.files else None
    c['symmetric'] = m['method'] == 'symmetric_uniform'
    c.update(LAYOUT_ARGS[m['layout']])
    return c


def est_inputs(z, m):
    name = m['inputs']
    src = {'batches': 'batches', 'wbatches': 'wbatches', 'logits': 'logits',
           'pos_batches': 'batches', 'wbig': 'wbig', 'abig': 'abig'}[name]
    xs = [t(b) for b in z[src]]
    if name == 'pos_batches':
        xs = [b.abs() for b in xs]
    return xs
