"""AdaRound options, annealing schedule, combined loss and layer I/O capture.

Counterpart of the reference's quantization/adaround/utils.py (enums :37-90, ``TempDecay``
:93-128, ``CombinedLoss`` :131-172, hooks + ``GetLayerInpOut`` :175-233, ``LayerOutputMSE``
:236-252).  The schedule is scalar host math; the tensor work (reconstruction loss, regulariser)
goes through ``tq_recon_loss`` / ``tq_adaround_reg`` in the fused optimisation loop
(quantization/adaround/adaround.py) and through differentiable torch ops when an external
optimizer drives ``CombinedLoss`` itself.
"""
import logging
from enum import Flag
from math import ceil

import numpy as np
import torch
import torch.nn.functional as F

from utils.utils import StopForwardException

logger = logging.getLogger('AdaRound')
logger.setLevel(logging.INFO)


def sigmoid(x):
    return (1.0 + np.exp(-x)) ** -1.0


class BaseOption(Flag):
    """Flag enum whose members print as their bare name (CLI-friendly)."""

    def __str__(self):
        return self.name

    @property
    def cls(self):
        return self.value.cls

    @classmethod
    def list_names(cls):
        return [m.name for m in cls]


class _RoundingOption(BaseOption):
    @classmethod
    def list_names(cls):
        # user-selectable relaxations only: neither the default nor the composite alias
        return [m.name for m in cls if m.name not in ('nearest', 'RELAXATION')]


def _flags(base, name, *members, **aliases):
    """Flag enum with power-of-two members in the given order (same values as `auto()`) plus
    composite aliases given as tuples of member names."""
    table = {m: 1 << i for i, m in enumerate(members)}
    for alias, parts in aliases.items():
        table[alias] = sum(table[p] for p in parts)
    return base(name, table, module=__name__)


# activations stay FP32 during AdaRound | are quantized afterwards (default)
AdaRoundActQuantMode = _flags(BaseOption, 'AdaRoundActQuantMode', 'no_act_quant', 'post_adaround')
# how the weight quantization grid is initialised
AdaRoundInitMode = _flags(BaseOption, 'AdaRoundInitMode', 'range_estimator', 'mse', 'mse_out',
                          'mse_out_asym')
# regularisation term
AdaRoundLossType = _flags(BaseOption, 'AdaRoundLossType', 'relaxation', 'temp_decay')
# rounding: nearest (default) or one of the continuous relaxations of the AdaRound paper
AdaRoundMode = _flags(_RoundingOption, 'AdaRoundMode', 'nearest', 'learned_sigmoid',
                      'learned_hard_sigmoid', 'sigmoid_temp_decay',
                      RELAXATION=('learned_sigmoid', 'learned_hard_sigmoid', 'sigmoid_temp_decay'))
AdaRoundTempDecayType = _flags(BaseOption, 'AdaRoundTempDecayType', 'linear', 'cosine', 'sigmoid',
                               'power', 'exp', 'log')

MODE_TO_LOSS_TYPE = {
    AdaRoundMode.learned_hard_sigmoid: AdaRoundLossType.relaxation,
    AdaRoundMode.learned_sigmoid: AdaRoundLossType.relaxation,
    AdaRoundMode.sigmoid_temp_decay: AdaRoundLossType.temp_decay,
}


class TempDecay:
    """beta(t): constant `start_b` until `rel_decay_start * t_max`, then annealed to `end_b`."""

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

        hi, lo, k = self.start_b, self.end_b, self.decay_shape
        rel_t = (t - self.decay_start) / (self.t_max - self.decay_start)
        kind = self.decay_type
        if kind == AdaRoundTempDecayType.linear:
            return lo + (hi - lo) * max(0.0, (1 - rel_t))
        if kind == AdaRoundTempDecayType.cosine:
            return lo + 0.5 * (hi - lo) * (1 + np.cos(rel_t * np.pi))
        if kind == AdaRoundTempDecayType.sigmoid:
            offset = sigmoid(-k / 2)
            progress = (sigmoid(k * (rel_t - 0.5)) - offset) / (1 - 2 * offset)
            return hi + (lo - hi) * progress
        if kind == AdaRoundTempDecayType.power:
            return lo + (hi - lo) * (1 - rel_t ** k)
        if kind == AdaRoundTempDecayType.exp:
            progress = (1.0 - np.exp(-k * rel_t)) / (1.0 - np.exp(-k))
            return hi + (lo - hi) * progress
        if kind == AdaRoundTempDecayType.log:
            c_end, c_start = np.exp(lo / k), np.exp(hi / k)
            return k * np.log((c_end - c_start) * rel_t + c_start)
        raise ValueError(f'Unknown temp decay type {self.decay_type}')


class CombinedLoss:
    """Reconstruction MSE + annealed rounding regulariser lambda * sum(1 - |2h - 1|^beta).

    ``schedule(it)`` exposes (beta, regulariser_active) for a 1-based iteration so the fused
    loop can feed them to ``tq_adaround_bwd_adam``; ``__call__`` is the differentiable version
    for callers that run their own autograd + optimizer."""

    def __init__(self, quantizer, loss_type=AdaRoundLossType.relaxation, weight=0.01,
                 max_count=1000, b_range=(20, 2), warmup=0.0, decay_start=0.0, **temp_decay_kw):
        self.quantizer = quantizer
        self.loss_type = loss_type
        self.weight = weight

        self.loss_start = max_count * warmup
        self.temp_decay = TempDecay(max_count, b_range=b_range,
                                    rel_decay_start=warmup + (1.0 - warmup) * decay_start,
                                    **temp_decay_kw)
        self.iter = 0

    def schedule(self, it):
        b = self.temp_decay(it)
        active = it >= self.loss_start and self.loss_type == AdaRoundLossType.relaxation
        return b, active

    def __call__(self, pred, tgt, *args, **kwargs):
        self.iter += 1
        rec_loss = F.mse_loss(pred, tgt, reduction='none').sum(1).mean()
        b = self.temp_decay(self.iter)

        round_loss = 0
        if self.iter < self.loss_start:
            pass
        elif self.loss_type == AdaRoundLossType.temp_decay:
            self.quantizer.temperature = b
        elif self.loss_type == AdaRoundLossType.relaxation:
            h = self.quantizer.get_rest().view(-1)
            round_loss = self.weight * (1 - ((h - 0.5).abs() * 2).pow(b)).sum()
        else:
            raise ValueError(f'Unknown loss type {self.loss_type}')

        total_loss = rec_loss + round_loss
        if self.iter == 1 or self.iter % 100 == 0:
            logger.info(f'Total loss:\t{total_loss:.4f} (rec:{rec_loss:.4f}, '
                        f'round:{round_loss:.3f})\tb={b:.2f}\titer={self.iter}')
        return total_loss


class StopForwardHook:
    def __call__(self, module, *args):
        raise StopForwardException


class DataSaverHook:
    """Forward hook that keeps the hooked layer's input and/or output."""

    def __init__(self, store_input=False, store_output=False, stop_forward=False):
        self.store_input = store_input
        self.store_output = store_output
        self.stop_forward = stop_forward
        self.input_store = None
        self.output_store = None

    def __call__(self, module, input_batch, output_batch):
        if self.store_input:
            self.input_store = input_batch
        if self.store_output:
            self.output_store = output_batch
        if self.stop_forward:
            raise StopForwardException


class GetLayerInpOut:
    """(input, FP32 output) of `layer` for a model input.  With `asym` the input is re-captured
    from a second pass through the weight-quantized network (so layer k sees the rounding already
    learned for layers < k); both passes stop at the layer."""

    def __init__(self, model, layer, asym=False, act_quant=False, store_output=True):
        self.model = model
        self.layer = layer
        self.asym = asym
        self.device = layer.weight.device
        self.act_quant = act_quant
        self.store_output = store_output
        self.data_saver = DataSaverHook(store_input=True, store_output=self.store_output,
                                        stop_forward=True)

    def _run_until_layer(self, model_input):
        try:
            self.model(model_input.to(self.device))
        except StopForwardException:
            pass

    def __call__(self, model_input):
        self.model.full_precision()
        handle = self.layer.register_forward_hook(self.data_saver)
        with torch.no_grad():
            self._run_until_layer(model_input)
            if self.asym:
                self.data_saver.store_output = False
                self.model.set_quant_state(weight_quant=True, act_quant=self.act_quant)
                self._run_until_layer(model_input)
                self.data_saver.store_output = True
        handle.remove()

        self.model.full_precision()
        self.layer.quantized_weights()
        return self.data_saver.input_store[0].detach(), self.data_saver.output_store.detach()


class LayerOutputMSE:
    """Sum over mini-batches of mse(layer(x), fp32 output) -- the score of the mse_out grid init."""

    def __init__(self, layer, get_inp_out, data_tensor, batch_size, name='mse_out'):
        self.input, self.exp_out = get_inp_out(data_tensor)
        self.layer = layer
        self.batch_size = batch_size
        self.name = name

    def __call__(self):
        loss = 0.0
        bs = self.batch_size
        for i in range(ceil(self.input.size(0) / bs)):
            cur_out = self.layer(self.input[i * bs:(i + 1) * bs])
            loss += F.mse_loss(cur_out, self.exp_out[i * bs:(i + 1) * bs]).item()
        return loss
