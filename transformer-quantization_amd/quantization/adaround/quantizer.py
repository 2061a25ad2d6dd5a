"""AdaRound weight quantizers: floor + learned rounding variable, as gfx950 kernels.

Counterpart of the reference's quantization/adaround/quantizer.py (``logit / hard_sigmoid /
hard_logit`` :22-34, ``AdaRoundQuantizer`` :37-100, ``ADAROUND_QUANTIZER_MAP`` :111-114).

    x_int = clamp(floor(w / s) + r (+ zero_point), int_min, int_max),   y = s * (x_int - zero_point)
    r     = h(alpha)  while optimising (soft_targets)   |   [alpha >= 0]  afterwards

The whole chain is ``tq_adaround_fwd`` (K10); alpha is initialised by ``tq_adaround_init_alpha``;
the gradient w.r.t. alpha comes from ``tq_adaround_bwd`` (or fused with Adam, K11).
"""
import logging

import torch
import torch.nn as nn
from torch.autograd import Function

from quantization import _hip
from quantization.adaround.utils import AdaRoundMode
from quantization.quantizers import (
    QuantizerBase,
    AsymmetricUniformQuantizer,
    SymmetricUniformQuantizer,
    param_layout,
)

logger = logging.getLogger('AdaRound')
logger.setLevel(logging.INFO)

_MODE_CODE = {
    AdaRoundMode.learned_sigmoid: _hip.ADA_SIGMOID,
    AdaRoundMode.learned_hard_sigmoid: _hip.ADA_HARD_SIGMOID,
    AdaRoundMode.sigmoid_temp_decay: _hip.ADA_SIGMOID_TEMP,
}


def logit(p, eps=1e-16):
    p = torch.clamp(p, eps, 1 - eps)
    return -torch.log(1 / p - 1)


def hard_sigmoid(x, zeta=1.1, gamma=-0.1):
    return torch.clamp(torch.sigmoid(x) * (zeta - gamma) + gamma, 0.0, 1.0)


def hard_logit(p, zeta=1.1, gamma=-0.1):
    # the argument of log stays within [1/11, 11] for the default stretch
    return -torch.log((zeta - p) / (p - gamma))


class _AdaRoundFn(Function):
    """w_q(w, alpha) through K10; d w_q / d alpha through tq_adaround_bwd (w gets no gradient:
    floor has zero derivative, as in the reference)."""

    @staticmethod
    def forward(ctx, w, alpha, quantizer, qargs, mode, soft, temperature):
        ctx.save_for_backward(w, alpha)
        ctx.meta = (qargs, mode, soft, temperature)
        return _hip.backend().adaround_fwd(w, alpha, qargs, mode, soft, temperature)

    @staticmethod
    def backward(ctx, grad_wq):
        w, alpha = ctx.saved_tensors
        qargs, mode, soft, temperature = ctx.meta
        if not soft or not ctx.needs_input_grad[1]:
            return None, None, None, None, None, None, None
        g = _hip.backend().adaround_bwd(w, alpha, grad_wq, qargs, mode, temperature)
        return None, g, None, None, None, None, None


class AdaRoundQuantizer(QuantizerBase):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.alpha = None
        self.round_mode = AdaRoundMode.nearest
        self.soft_targets = False
        self.temperature = None  # sigmoid temperature annealing

    # ---- kernel plumbing ---------------------------------------------------------------
    def _relaxed(self):
        if self.round_mode == AdaRoundMode.nearest:
            return False
        if self.round_mode not in AdaRoundMode.RELAXATION:
            raise ValueError(f'Unknown rounding mode: {self.round_mode}')
        return True

    def kernel_args(self, w):
        """The tq_quantizer fields for a weight tensor `w` (see _hip.HipBackend._qdesc)."""
        n_params, inner = param_layout(w, self.delta.numel(), None, self.per_channel,
                                       self.delta.shape)
        return (self.delta, self._zero_float, getattr(self, '_signed', None), self.n_bits,
                self.symmetric, self.scale_domain == 'log', self.eps, n_params, inner)

    def mode_code(self):
        try:
            return _MODE_CODE[self.round_mode]
        except KeyError:
            raise ValueError(f'Unknown rounding mode: {self.round_mode}') from None

    def _ensure_alpha(self, w):
        if self.alpha is None:
            logger.info('Init alpha to be FP32')
            alpha = _hip.backend().adaround_init_alpha(w, self.kernel_args(w), self.mode_code(),
                                                       self.temperature)
            self.alpha = nn.Parameter(alpha, requires_grad=True)

    # ---- reference API -----------------------------------------------------------------
    def to_integer_forward(self, x_float):
        if not self._relaxed():
            return super().to_integer_forward(x_float)
        self._ensure_alpha(x_float)
        zp = 0.0 if self.symmetric else self.zero_point
        if not self.soft_targets:
            # hard targets: floor(x / s) + [alpha >= 0] (+ zp), clamped -- an exact integer.  Recovered from the kernel's
            # dequantised output s * (x_int - zp): the quotient is within one ulp of that integer (|x_int - zp| < 2^22),
            # so rounding restores it exactly (reference adaround/quantizer.py:72-79 returns the integer itself;
            # found by tests/test_adaround_model.py: y / s alone gave 2.9999998 for 134 of 1152 entries of a layer).
            y = self.forward(x_float)
            return torch.round(y / self.scale) + zp
        # soft targets: x_floor + h(alpha), differentiable in alpha -- the reference's own expression
        # (adaround/quantizer.py:54-79); not on any hot path (the optimisation loop uses the fused kernels)
        x_floor = torch.floor(x_float / self.scale)
        x_int = x_floor + self.get_rest()
        if not self.symmetric:
            x_int = x_int + zp
        return torch.clamp(x_int, self.int_min, self.int_max)

    def forward(self, x_float):
        if not self._relaxed():
            return super().forward(x_float)
        if self.per_channel:
            self._adjust_params_per_channel(x_float)
        self._ensure_alpha(x_float)
        args = (x_float, self.alpha, self, self.kernel_args(x_float), self.mode_code(),
                bool(self.soft_targets), self.temperature)
        if torch.is_grad_enabled() and self.alpha.requires_grad and self.soft_targets:
            return _AdaRoundFn.apply(*args)
        return _hip.backend().adaround_fwd(x_float, self.alpha.detach(), *args[3:])

    def get_rest(self):
        """h(alpha) as a differentiable torch expression (used by CombinedLoss.__call__)."""
        if self.round_mode == AdaRoundMode.learned_sigmoid:
            return torch.sigmoid(self.alpha)
        if self.round_mode == AdaRoundMode.learned_hard_sigmoid:
            return hard_sigmoid(self.alpha)
        if self.round_mode == AdaRoundMode.sigmoid_temp_decay:
            return torch.sigmoid(self.alpha / self.temperature)
        raise ValueError(f'Unknown rounding mode: {self.round_mode}')

    def extra_repr(self):
        return ', '.join([
            f'n_bits={self.n_bits}',
            f'per_channel={self.per_channel}',
            f'is_initialized={self.is_initialized}',
            f'round_mode={self.round_mode}',
            f'soft_targets={self.soft_targets}',
            f'temperature={self.temperature}',
        ])


class AdaRoundSymmetricUniformQuantizer(AdaRoundQuantizer, SymmetricUniformQuantizer):
    pass


class AdaRoundAsymmetricUniformQuantizer(AdaRoundQuantizer, AsymmetricUniformQuantizer):
    pass


ADAROUND_QUANTIZER_MAP = {
    SymmetricUniformQuantizer: AdaRoundSymmetricUniformQuantizer,
    AsymmetricUniformQuantizer: AdaRoundAsymmetricUniformQuantizer,
}
