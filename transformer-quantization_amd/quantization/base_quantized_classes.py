"""Module-level building blocks: QuantizedModule, QuantizedActivation, FP32Acts.

Counterpart of the reference's ``quantization/base_quantized_classes.py`` (:35-155); pure host
logic (flags, caches, state broadcast through ``nn.Module.apply``) -- the numerics are in the
``QuantizationManager`` each block owns.
"""
from torch import nn

from quantization.quantization_manager import QuantizationManager
from quantization.quantizers import QMethods
from quantization.range_estimators import RangeEstimators


def _if_initialised(action):
    """Visitor for nn.Module.apply: call `action` on every manager that already holds a range."""
    def visit(layer):
        if isinstance(layer, QuantizationManager) and layer.quantizer.is_initialized:
            getattr(layer, action)()
    visit.__name__ = '_set_layer_' + action
    return visit


_set_layer_learn_ranges = _if_initialised('learn_ranges')
_set_layer_fix_ranges = _if_initialised('fix_ranges')
_set_layer_estimate_ranges = _if_initialised('estimate_ranges')
_set_layer_estimate_ranges_train = _if_initialised('estimate_ranges_train')


class QuantizedModule(nn.Module):
    """Per-layer quantization configuration + on/off switches + the cache of quantized parameters.

    Keyword arguments (the reference's whole `qparams` surface, base_quantized_classes.py:41-60):
    method / act_method (QMethods), n_bits / n_bits_act, per_channel_weights / per_channel_acts,
    percentile, weight_range_method / weight_range_options, act_range_method / act_range_options
    (RangeEstimators + their kwargs), scale_domain.  ``quant_dict`` is accepted and ignored here.
    """

    def __init__(self, *args, method=QMethods.asymmetric_uniform, act_method=None, n_bits=8,
                 n_bits_act=None, per_channel_weights=False, per_channel_acts=False,
                 percentile=None, weight_range_method=RangeEstimators.current_minmax,
                 weight_range_options=None, act_range_method=RangeEstimators.running_minmax,
                 act_range_options=None, scale_domain='linear', **kwargs):
        kwargs.pop('quant_dict', None)
        super().__init__(*args, **kwargs)

        # weights
        self.method, self.n_bits = method, n_bits
        self.per_channel_weights, self.percentile = per_channel_weights, percentile
        self.weight_range_method = weight_range_method
        self.weight_range_options = dict(weight_range_options) if weight_range_options else {}
        # activations default to the weight settings
        self.act_method = act_method if act_method else method
        self.n_bits_act = n_bits_act if n_bits_act else n_bits
        self.per_channel_acts = per_channel_acts
        self.act_range_method = act_range_method
        self.act_range_options = dict(act_range_options) if act_range_options else {}
        self.scale_domain = scale_domain

        self.quant_params = None
        self.cached_params, self._caching = None, True
        self._quant_w = self._quant_a = False

    # ---- quantized-parameter cache: dropped whenever it could be stale ---------------------------
    @property
    def caching(self):
        return self._caching

    @caching.setter
    def caching(self, value: bool):
        self._caching = value
        if not value:
            self.cached_params = None

    def train(self, mode=True):
        if mode:
            self.cached_params = None
        return super().train(mode)

    def _apply(self, *args, **kwargs):      # .to() / .cuda() / .float() move the parameters
        self.cached_params = None
        return super()._apply(*args, **kwargs)

    # ---- on / off ---------------------------------------------------------------------------------
    def _weights(self, on):
        self.cached_params = None
        self._quant_w = on

    def quantized_weights(self):
        self._weights(True)

    def full_precision_weights(self):
        self._weights(False)

    def quantized_acts(self):
        self._quant_a = True

    def full_precision_acts(self):
        self._quant_a = False

    def quantized(self):
        self._weights(True)
        self._quant_a = True

    def full_precision(self):
        self._weights(False)
        self._quant_a = False

    # ---- range states of every manager below this module ------------------------------------------
    def learn_ranges(self):
        self.apply(_set_layer_learn_ranges)

    def fix_ranges(self):
        self.apply(_set_layer_fix_ranges)

    def estimate_ranges(self):
        self.apply(_set_layer_estimate_ranges)

    def estimate_ranges_train(self):
        self.apply(_set_layer_estimate_ranges_train)

    def extra_repr(self):
        own = f'weight_quant={self._quant_w}, act_quant={self._quant_a}'
        inherited = super().extra_repr()
        return f'{inherited},\n{own}' if inherited else own


class QuantizedActivation(QuantizedModule):
    """Stand-alone activation quantizer (residual sums, attention scores / probs, embedding sums)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.activation_quantizer = QuantizationManager(
            qmethod=self.act_method, init=self.act_range_method,
            qparams={'n_bits': self.n_bits_act, 'scale_domain': self.scale_domain},
            init_params=self.act_range_options)

    def quantize_activations(self, x):
        # (sub-module through the registry: nn.Module.__getattr__ is the slow path of attribute access, and a fixed-range
        # call is launch-bound)
        if not self._quant_a:
            return x
        m = self._modules['activation_quantizer']
        return m.quantize(x) if type(m) is QuantizationManager else m(x)

    def forward(self, x):
        return self.quantize_activations(x)


class FP32Acts(nn.Module):
    """Identity stand-in for a disabled quantizer."""

    def forward(self, x):
        return x

    def reset_ranges(self):
        pass
