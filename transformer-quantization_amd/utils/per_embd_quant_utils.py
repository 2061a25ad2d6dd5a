"""Per-embedding / per-embedding-group (PEG) and mixed-precision re-targeting of existing
quantizers (counterpart of the reference's utils/per_embd_quant_utils.py :7-68).

quant_dict values: int -> bit-width, 'fp32' -> disable, 'per_embd' -> one range per embedding
dimension (axis 2), 'ngK' -> K groups, 'ngpK' -> K groups after sorting dimensions by range.
"""
from quantization.base_quantized_classes import FP32Acts


def set_act_quant_axis_and_groups(module, axis, n_groups, permute=False):
    mgr = module.activation_quantizer if hasattr(module, 'activation_quantizer') else module

    mgr.axis = axis
    mgr.quantizer.axis = axis
    mgr.range_estimator.axis = axis

    mgr.n_groups = n_groups
    mgr.range_estimator.n_groups = n_groups

    if permute:
        mgr.range_estimator.per_group_range_estimation = True
    return mgr


def _hijack_act_quant(module, value):
    if value is None:
        return
    if isinstance(value, int):
        module.activation_quantizer.quantizer.n_bits = value
    elif value == 'fp32':
        module.activation_quantizer = FP32Acts()
    elif value == 'per_embd':
        set_act_quant_axis_and_groups(module, axis=2, n_groups=None)
    elif value.startswith('ngp'):
        set_act_quant_axis_and_groups(module, axis=2, n_groups=int(value[3:]), permute=True)
    elif value.startswith('ng'):
        set_act_quant_axis_and_groups(module, axis=2, n_groups=int(value[2:]), permute=False)
    else:
        raise NotImplementedError(f'Unknown value "{value}" in quant_dict')


def _hijack_weight_quant(module, value):
    if value is None:
        return
    if isinstance(value, int):
        module.weight_quantizer.quantizer.n_bits = value
    elif value == 'fp32':
        module.weight_quantizer = FP32Acts()
    else:
        raise NotImplementedError(f'Unknown value "{value}" in quant_dict')


def hijack_act_quant(quant_dict, name, m):
    _hijack_act_quant(m, quant_dict.get(name, None))


def hijack_weight_quant(quant_dict, name, m):
    _hijack_weight_quant(m, quant_dict.get(name, None))


def hijack_act_quant_modules(quant_dict, name, m):
    value = quant_dict.get(name, None)
    for sub in m.modules():
        if hasattr(sub, 'activation_quantizer'):
            _hijack_act_quant(sub, value)
