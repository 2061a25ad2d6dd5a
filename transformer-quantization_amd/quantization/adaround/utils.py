"""AdaRound options, annealing schedule, losses and layer I/O capture for the MI355X path.

API counterpart of the reference's quantization/adaround/utils.py (option enums :37-90, ``TempDecay`` :93-128,
``CombinedLoss`` :131-172, ``GetLayerInpOut`` :198-233, ``LayerOutputMSE`` :236-252): names, constructor arguments
and the values they produce are the contract; the structure is built around the device path:

* every loss VALUE lives on the device until somebody prints it: ``LayerOutputMSE`` (the score of the ``mse_out``
  grid inits) adds one fp64 cell per mini-batch through ``tq_recon_loss`` and returns a 0-d device tensor -- the
  reference calls ``.item()`` per mini-batch, i.e. 80 candidates x #batches host synchronisations per layer;
* ``CombinedLoss`` is a schedule (``schedule(it)`` -> beta, regulariser on/off) consumed by the fused step
  ``tq_adaround_bwd_adam`` (quantization/adaround/adaround.py); its ``__call__`` is the differentiable form for
  callers that run their own autograd + optimizer (activation quantizer on, external optimizers);
* layer I/O capture keeps everything in HBM (288 GB: the cached [1024, 128, 3072] fp32 outputs of BERT-base's
  widest layer are 1.6 GB).
"""
import logging
from contextlib import contextmanager
from enum import Flag, auto

import numpy as np
import torch
import torch.nn.functional as F

from quantization import _hip
from utils.utils import StopForwardException

logger = logging.getLogger('AdaRound')
logger.setLevel(logging.INFO)


# ------------------------------------------------------------------------------------------- options
class BaseOption(Flag):
    def __str__(self):
        return self.name

    @property
    def cls(self):
        return self.value.cls

    @classmethod
    def list_names(cls):
        return [m.name for m in cls]


class AdaRoundActQuantMode(BaseOption):
    no_act_quant = auto()       # activations stay FP32
    post_adaround = auto()      # activations are quantized after AdaRound (default)


class AdaRoundInitMode(BaseOption):
    range_estimator = auto()    # grid from the weight range estimator
    mse = auto()                # grid minimising the weight MSE
    mse_out = auto()            # grid minimising the layer-output MSE
    mse_out_asym = auto()       # same, inputs taken from the already quantized network


class AdaRoundMode(BaseOption):
    nearest = auto()
    learned_sigmoid = auto()
    learned_hard_sigmoid = auto()
    sigmoid_temp_decay = auto()
    RELAXATION = learned_sigmoid | learned_hard_sigmoid | sigmoid_temp_decay

    @classmethod
    def list_names(cls):
        # the user-selectable relaxations: neither the default nor the composite alias
        return [m.name for m in cls if m not in (cls.nearest, cls.RELAXATION)]


class AdaRoundTempDecayType(BaseOption):
    linear = auto()
    cosine = auto()
    sigmoid = auto()
    power = auto()
    exp = auto()
    log = auto()


class AdaRoundLossType(BaseOption):
    relaxation = auto()
    temp_decay = auto()


MODE_TO_LOSS_TYPE = {
    AdaRoundMode.learned_hard_sigmoid: AdaRoundLossType.relaxation,
    AdaRoundMode.learned_sigmoid: AdaRoundLossType.relaxation,
    AdaRoundMode.sigmoid_temp_decay: AdaRoundLossType.temp_decay,
}


# ------------------------------------------------------------------------------------------- schedule
def _logistic(v):
    return (1.0 + np.exp(-v)) ** -1.0


# beta(start, end, rel_t, shape) of every annealing curve.  Same floating-point expressions as upstream (:111-128):
# the schedule feeds the regulariser exponent, and the alpha traces of the parity fixtures depend on its exact value
# (numpy's exp / cos / log, not libm's: they differ in the last bit).
_CURVES = {
    'linear': lambda s, e, r, k: e + (s - e) * max(0.0, (1 - r)),
    'cosine': lambda s, e, r, k: e + 0.5 * (s - e) * (1 + np.cos(r * np.pi)),
    'sigmoid': lambda s, e, r, k: s + (e - s) * ((_logistic(k * (r - 0.5)) - _logistic(-k / 2)) /
                                                 (1 - 2 * _logistic(-k / 2))),
    'power': lambda s, e, r, k: e + (s - e) * (1 - r ** k),
    'exp': lambda s, e, r, k: s + (e - s) * ((1.0 - np.exp(-k * r)) / (1.0 - np.exp(-k))),
    'log': lambda s, e, r, k: k * np.log((np.exp(e / k) - np.exp(s / k)) * r + np.exp(s / k)),
}


class TempDecay:
    """beta(t): `b_range[0]` until `rel_decay_start * t_max`, then annealed to `b_range[1]` along `decay_type`."""

    def __init__(self, t_max, b_range=(20.0, 2.0), rel_decay_start=0.0,
                 decay_type=AdaRoundTempDecayType.linear, decay_shape=1.0):
        self.t_max = t_max
        self.start_b, self.end_b = b_range
        self.decay_type = decay_type
        self.decay_shape = decay_shape
        self.decay_start = rel_decay_start * t_max

    def __call__(self, t):
        if t < self.decay_start:
            return self.start_b
        curve = _CURVES.get(getattr(self.decay_type, 'name', None))
        if curve is None:
            raise ValueError(f'Unknown temp decay type {self.decay_type}')
        rel_t = (t - self.decay_start) / (self.t_max - self.decay_start)
        return curve(self.start_b, self.end_b, rel_t, self.decay_shape)


# ------------------------------------------------------------------------------------------- losses
class CombinedLoss:
    """Reconstruction error  mse(pred, tgt, 'none').sum(1).mean()  +  lambda * sum(1 - |2 h(alpha) - 1| ** beta(t)).

    The fused optimisation step asks `schedule(it)` for (beta, regulariser active) and leaves the arithmetic to
    `tq_adaround_bwd_adam`; `value(...)` evaluates the same total on the device for logging (`tq_recon_loss`,
    `tq_adaround_reg`); calling the object gives the differentiable torch expression."""

    def __init__(self, quantizer, loss_type=AdaRoundLossType.relaxation, weight=0.01, max_count=1000,
                 b_range=(20, 2), warmup=0.0, decay_start=0.0, **temp_decay_kw):
        self.quantizer = quantizer
        self.loss_type = loss_type
        self.weight = weight
        self.loss_start = max_count * warmup
        self.temp_decay = TempDecay(max_count, b_range=b_range,
                                    rel_decay_start=warmup + (1.0 - warmup) * decay_start, **temp_decay_kw)
        self.iter = 0

    def schedule(self, it):
        """(beta, regulariser active) of the 1-based iteration `it`."""
        return self.temp_decay(it), (it >= self.loss_start and self.loss_type == AdaRoundLossType.relaxation)

    def value(self, pred, tgt, it, scale=1.0):
        """(reconstruction, rounding) terms as 0-d fp64 device tensors, no host synchronisation."""
        be = _hip.backend()
        q = self.quantizer
        b, active = self.schedule(it)
        rec = be.recon_loss(pred, tgt) * scale
        rnd = be.adaround_reg(q.alpha, q.mode_code(), q.temperature, b, self.weight) if active else torch.zeros_like(rec)
        return rec, rnd

    def __call__(self, pred, tgt, *args, **kwargs):
        self.iter += 1
        b = self.temp_decay(self.iter)
        total = F.mse_loss(pred, tgt, reduction='none').sum(1).mean()
        rec, rnd = total, 0.0
        if self.iter >= self.loss_start:
            if self.loss_type == AdaRoundLossType.temp_decay:
                self.quantizer.temperature = b
            elif self.loss_type == AdaRoundLossType.relaxation:
                h = self.quantizer.get_rest().reshape(-1)
                rnd = self.weight * (1 - ((h - 0.5).abs() * 2).pow(b)).sum()
                total = rec + rnd
            else:
                raise ValueError(f'Unknown loss type {self.loss_type}')
        if self.iter == 1 or self.iter % 100 == 0:
            logger.info(f'Total loss:\t{total:.4f} (rec:{rec:.4f}, round:{rnd:.3f})\tb={b:.2f}\titer={self.iter}')
        return total


class LayerOutputMSE:
    """Score of the `mse_out` grid inits: sum over mini-batches of mse_loss(layer(x_b), fp32_out_b).

    Returns a 0-d fp64 DEVICE tensor: per mini-batch one `tq_recon_loss` cell (the mean squared error of the batch)
    added on the device; the caller compares / reduces scores there and synchronises once."""

    def __init__(self, layer, get_inp_out, data_tensor, batch_size, name='mse_out'):
        self.input, self.exp_out = get_inp_out(data_tensor)
        self.layer = layer
        self.batch_size = batch_size
        self.name = name

    def __call__(self):
        be = _hip.backend()
        total = None
        for lo in range(0, self.input.size(0), self.batch_size):
            out = self.layer(self.input[lo:lo + self.batch_size])
            # plain mean over all elements == F.mse_loss(out, expected)
            cell = be.recon_loss(out.reshape(-1, 1), self.exp_out[lo:lo + self.batch_size].reshape(-1, 1))
            total = cell if total is None else total + cell
        return total


# ------------------------------------------------------------------------------------------- layer I/O
class StopForwardHook:
    def __call__(self, module, *args):
        raise StopForwardException


class DataSaverHook:
    """Forward hook that keeps (references to) the hooked layer's input / output and can abort the forward."""

    def __init__(self, store_input=False, store_output=False, stop_forward=False):
        self.store_input, self.store_output, self.stop_forward = store_input, store_output, stop_forward
        self.input_store = self.output_store = None

    def __call__(self, module, input_batch, output_batch):
        if self.store_input:
            self.input_store = input_batch
        if self.store_output:
            self.output_store = output_batch
        if self.stop_forward:
            raise StopForwardException


@contextmanager
def _hooked(layer, hook):
    handle = layer.register_forward_hook(hook)
    try:
        yield hook
    finally:
        handle.remove()


class GetLayerInpOut:
    """model input -> (input of `layer`, FP32 output of `layer`), both forwards aborted at the layer.

    `asym`: the input is taken from a second pass through the network with quantized weights (and activations if
    `act_quant`), so layer k is optimised on what it will really see after the rounding of layers < k."""

    def __init__(self, model, layer, asym=False, act_quant=False, store_output=True):
        self.model, self.layer = model, layer
        self.asym, self.act_quant, self.store_output = asym, act_quant, store_output
        self.device = layer.weight.device
        self.data_saver = DataSaverHook(store_input=True, store_output=store_output, stop_forward=True)

    def _forward_to_layer(self, model_input):
        try:
            self.model(model_input.to(self.device))
        except StopForwardException:
            pass

    def __call__(self, model_input):
        saver = self.data_saver
        self.model.full_precision()
        with _hooked(self.layer, saver), torch.no_grad():
            self._forward_to_layer(model_input)
            if self.asym:
                saver.store_output = False                   # keep the FP32 output, replace the input
                self.model.set_quant_state(weight_quant=True, act_quant=self.act_quant)
                self._forward_to_layer(model_input)
                saver.store_output = self.store_output
        self.model.full_precision()
        self.layer.quantized_weights()
        return saver.input_store[0].detach(), saver.output_store.detach()
