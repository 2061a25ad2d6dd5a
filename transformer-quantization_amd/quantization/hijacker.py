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

        self.activation_quantizer = self._manager(
            self.act_method, self.act_range_method, self.per_channel_acts, self.n_bits_act,
            self.act_range_options)
        # `percentile` is understood by the current-min/max estimator only
        weight_options = ({'percentile': self.percentile}
                          if self.weight_range_method == RangeEstimators.current_minmax
                          else self.weight_range_options)
        self.weight_quantizer = self._manager(
            self.method, self.weight_range_method, self.per_channel_weights, self.n_bits,
            weight_options)

        self.activation_save_target = None
        self.activation_save_name = None

    def _manager(self, qmethod, estimator, per_channel, n_bits, estimator_options):
        return QuantizationManager(
            qmethod=qmethod, init=estimator, per_channel=per_channel, init_params=estimator_options,
            qparams={'n_bits': n_bits, 'scale_domain': self.scale_domain})

    # ---- forward ------------------------------------------------------------------------------
    def forward(self, x, offsets=None):
        weight, bias = self.get_params()
        return self.quantize_activations(self.run_forward(x, weight, bias, offsets=offsets))

    def run_forward(self, x, weight, bias, offsets=None):
        """The wrapped layer's own computation; provided by the concrete class."""
        raise NotImplementedError()

    def get_weight_bias(self):
        return self.weight, getattr(self, 'bias', None)

    def get_params(self):
        """(weight, bias) with the weight fake-quantized if enabled; cached in eval mode."""
        cache_usable = not self.training
        if cache_usable and self.cached_params:
            return self.cached_params

        weight, bias = self.get_weight_bias()
        if self._quant_w:
            weight = self.weight_quantizer(weight)

        if cache_usable and self._caching and self.cached_params is None:
            # detached copies, like the reference's numpy round trip -- but resident in HBM.  Low-precision parameters
            # are widened to fp32 like `torch.Tensor(ndarray)` does; float64 (--double) is KEPT: the reference's
            # constructor narrows it to fp32 as well, which makes its second eval forward fail on a double x float
            # matmul (or silently continue in fp32 behind an embedding) -- not a behaviour worth reproducing.
            keep = weight.dtype if weight.dtype == torch.float64 else torch.float32
            self.cached_params = (weight.detach().to(keep), None if bias is None else bias.detach().to(keep))
        return weight, bias

    def _save(self, suffix, tensor):
        if self.activation_save_target is not None:
            self.activation_save_target[self.activation_save_name + suffix] = tensor.data.cpu().numpy()

    def quantize_activations(self, activations):
        """Optional activation function, then one output quantizer for the whole layer."""
        if self.activation_function is not None:
            activations = self.activation_function(activations)
        self._save('', activations)
        if self._quant_a:
            m = self._modules['activation_quantizer']
            activations = m.quantize(activations) if type(m) is QuantizationManager else m(activations)
            self._save('_Q', activations)
        return activations
