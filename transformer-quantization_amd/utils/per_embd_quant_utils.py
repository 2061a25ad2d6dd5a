"""Per-embedding / per-embedding-group (PEG) and mixed-precision re-targeting of existing
quantizers (counterpart of the reference's utils/per_embd_quant_utils.py :7-68).

quant_dict values: int -> bit-width, 'fp32' -> disable, 'per_embd' -> one range per embedding
dimension (axis 2), 'ngK' -> K groups, 'ngpK' -> K groups after sorting dimensions by range.
"""
import re

from quantization.base_quantized_classes import FP32Acts

_GROUPS = re.compile(r'^ng(p?)(\d+)$')


def set_act_quant_axis_and_groups(module, axis, n_groups, permute=False):
    """Point a manager (or the module owning it), its quantizer and its estimator at `axis`, with
    optional grouping; `permute` arms the range-collection phase of permuted PEG."""
    mgr = getattr(module, 'activation_quantizer', module)
    for target in (mgr, mgr.quantizer, mgr.range_estimator):
        target.axis = axis
    for target in (mgr, mgr.range_estimator):
        target.n_groups = n_groups
    if permute:
        mgr.range_estimator.per_group_range_estimation = True
    return mgr


def _unknown(value):
    raise NotImplementedError(f'Unknown value "{value}" in quant_dict')


def _hijack_act_quant(module, value):
    if value is None:
        return
    if isinstance(value, int):
        module.activation_quantizer.quantizer.n_bits = value
        return
    if value == 'fp32':
        module.activation_quantizer = FP32Acts()
        return
    if value == 'per_embd':
        set_act_quant_axis_and_groups(module, axis=2, n_groups=None)
        return
    m = _GROUPS.match(value) if isinstance(value, str) else None
    if m is None:
        _unknown(value)
    set_act_quant_axis_and_groups(module, axis=2, n_groups=int(m.group(2)), permute=bool(m.group(1)))


def _hijack_weight_quant(module, value):
    if value is None:
        return
    if isinstance(value, int):
        module.weight_quantizer.quantizer.n_bits = value
    elif value == 'fp32':
        module.weight_quantizer = FP32Acts()
    else:
        _unknown(value)


def hijack_act_quant(quant_dict, name, m):
    _hijack_act_quant(m, quant_dict.get(name))


def hijack_weight_quant(quant_dict, name, m):
    _hijack_weight_quant(m, quant_dict.get(name))


def hijack_act_quant_modules(quant_dict, name, m):
    value = quant_dict.get(name)
    for sub in m.modules():
        if hasattr(sub, 'activation_quantizer'):
            _hijack_act_quant(sub, value)
