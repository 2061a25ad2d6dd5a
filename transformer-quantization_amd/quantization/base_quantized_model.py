"""Model-wide switches (counterpart of the reference's quantization/base_quantized_model.py)."""
from torch import nn

from quantization.base_quantized_classes import (
    QuantizedModule,
    _set_layer_learn_ranges,
    _set_layer_fix_ranges,
    _set_layer_estimate_ranges,
    _set_layer_estimate_ranges_train,
)


def _on_quantized_modules(method_name):
    def toggle(self):
        def visit(layer):
            if isinstance(layer, QuantizedModule):
                getattr(layer, method_name)()
        self.apply(visit)
    toggle.__name__ = method_name
    return toggle


def _on_managers(attr, action):
    def toggle(self):
        def visit(module):
            if isinstance(module, QuantizedModule) and hasattr(module, attr):
                action(getattr(module, attr))
        self.apply(visit)
    return toggle


class QuantizedModel(nn.Module):
    """Convenience parent for a quantized network: flips every QuantizedModule /
    QuantizationManager underneath it (reference :15-113)."""

    quantized_weights = _on_quantized_modules('quantized_weights')
    full_precision_weights = _on_quantized_modules('full_precision_weights')
    quantized_acts = _on_quantized_modules('quantized_acts')
    full_precision_acts = _on_quantized_modules('full_precision_acts')
    quantized = _on_quantized_modules('quantized')
    full_precision = _on_quantized_modules('full_precision')

    def learn_ranges(self):
        self.apply(_set_layer_learn_ranges)

    def fix_ranges(self):
        self.apply(_set_layer_fix_ranges)

    def estimate_ranges(self):
        self.apply(_set_layer_estimate_ranges)

    def estimate_ranges_train(self):
        self.apply(_set_layer_estimate_ranges_train)

    fix_act_ranges = _on_managers('activation_quantizer', _set_layer_fix_ranges)
    fix_weight_ranges = _on_managers('weight_quantizer', _set_layer_fix_ranges)
    estimate_act_ranges = _on_managers('activation_quantizer', _set_layer_estimate_ranges)
    reset_act_ranges = _on_managers('activation_quantizer', lambda m: m.reset_ranges())

    def set_quant_state(self, weight_quant, act_quant):
        (self.quantized_acts if act_quant else self.full_precision_acts)()
        (self.quantized_weights if weight_quant else self.full_precision_weights)()
