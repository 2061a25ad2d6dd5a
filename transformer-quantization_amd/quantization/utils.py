"""Small tensor helpers (counterpart of the reference's quantization/utils.py)."""
import numpy as np


def to_numpy(tensor):
    """Host numpy view/copy of a tensor-like value (device tensors are copied to the host)."""
    if isinstance(tensor, np.ndarray):
        return tensor
    if hasattr(tensor, 'detach'):
        t = tensor.detach()
        return (t.cpu() if getattr(t, 'is_cuda', False) else t).numpy()
    if hasattr(tensor, 'numpy'):
        return tensor.numpy()
    return np.array(tensor)
