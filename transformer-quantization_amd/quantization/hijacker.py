"""QuantizationHijacker: turns an nn.Linear / nn.LayerNorm / nn.Embedding into
"fake-quantize the weight -> run the op -> activation function -> fake-quantize the output".

Counterpart of the reference's ``quantization/hijacker.py`` (:18-116).  The weight and output
quantizers are ``QuantizationManager``s, so both fake-quant steps are single HIP launches; in eval
mode the quantized weight is cached on the device (the reference round-trips the cache through a
host numpy array, hijacker.py:82-85 -- here it never leaves HBM, but keeps the fp32 cast).
"""
import copy

import torch
from torch import nn

from quantization.base_quantized_classes import QuantizedModule
from quantization.quantization_manager import QuantizationManager
from quantization.range_estimators import RangeEstimators

activations_list = [nn.ReLU, nn.ReLU6, nn.Hardtanh, nn.Sigmoid, nn.Tanh, nn.PReLU, nn.GELU]


class QuantizationHijacker(QuantizedModule):
    """Mixin; must precede the nn.Module it hijacks in the MRO:

    >>> class QuantLinear(QuantizationHijacker, nn.Linear):
    ...     def run_forward(self, x, weight, bias, offsets=None): ...
    """

    def __init__(self, *args, activation: nn.Module = None, **kwargs):
        super().__init__(*args, **kwargs)
        if activation:
            assert isinstance(activation, tuple(activations_list))
        self.activation_function = copy.deepcopy(activation) if activation else None

        self.activation_quantizer = QuantizationManager(
            qmethod=self.act_method,
            init=self.act_range_method,
            per_channel=self.per_channel_acts,
            qparams=dict(n_bits=self.n_bits_act, scale_domain=self.scale_domain),
            init_params=self.act_range_options,
        )

        # current_minmax is the only weight estimator that understands `percentile`
        if self.weight_range_method == RangeEstimators.current_minmax:
            weight_init_params = dict(percentile=self.percentile)
        else:
            weight_init_params = self.weight_range_options
        self.weight_quantizer = QuantizationManager(
            qmethod=self.method,
            init=self.weight_range_method,
            per_channel=self.per_channel_weights,
            qparams=dict(n_bits=self.n_bits, scale_domain=self.scale_domain),
            init_params=weight_init_params,
        )
        self.activation_save_target = None
        self.activation_save_name = None

    def forward(self, x, offsets=None):
        weight, bias = self.get_params()
        out = self.run_forward(x, weight, bias, offsets=offsets)
        return self.quantize_activations(out)

    def get_params(self):
        if not self.training and self.cached_params:
            return self.cached_params

        weight, bias = self.get_weight_bias()
        if self._quant_w:
            weight = self.weight_quantizer(weight)

        if self._caching and not self.training and self.cached_params is None:
            self.cached_params = (
                weight.detach().to(torch.float32),
                bias.detach().to(torch.float32) if bias is not None else None,
            )
        return weight, bias

    def get_weight_bias(self):
        return self.weight, (self.bias if hasattr(self, 'bias') else None)

    def run_forward(self, x, weight, bias, offsets=None):
        """The wrapped layer's own computation; provided by the concrete class."""
        raise NotImplementedError()

    def quantize_activations(self, activations):
        """Optional activation function, then one output quantizer for the whole layer."""
        if self.activation_function is not None:
            activations = self.activation_function(activations)

        if self.activation_save_target is not None:
            self.activation_save_target[self.activation_save_name] = activations.data.cpu().numpy()

        if self._quant_a:
            activations = self.activation_quantizer(activations)
            if self.activation_save_target is not None:
                self.activation_save_target[self.activation_save_name + '_Q'] = \
                    activations.data.cpu().numpy()

        return activations
