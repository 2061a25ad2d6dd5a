"""Uniform affine fake-quantizers whose numerics run as gfx950 HIP kernels.

Drop-in for the reference's ``quantization/quantizers.py`` (class names, constructor arguments,
properties, registered buffers ``_delta`` / ``_zero_float`` / ``_signed`` and exceptions are the
same), but ``forward`` / ``to_integer_forward`` / ``set_quant_range`` are single launches of
``libtq_hip.so`` (``tq_fake_quant_fwd``, ``tq_set_range_*``) instead of chains of ATen ops:

* reference ``AsymmetricUniformQuantizer.forward``          quantizers.py:189-211  -> K1/K2
* reference ``to_integer_forward``                           quantizers.py:172-187  -> K3
* reference ``set_quant_range`` (asym / sym)                 quantizers.py:263-282, 334-344
* STE backward of round (``RoundStraightThrough``)           quantizers.py:12-19 -> ``tq_fake_quant_bwd``

The raw buffers stay on the device and the kernels derive scale / zero-point / grid limits
from them, so a forward needs no host synchronisation (the reference syncs in ``signed`` on
every ``int_min`` / ``int_max`` access, quantizers.py:310-328).
"""
from collections import namedtuple
from enum import Enum

import torch
from torch import nn
from torch.autograd import Function

from quantization import _hip, options


class RoundStraightThrough(Function):
    """round-half-to-even forward, identity backward (reference quantizers.py:12-19)."""

    @staticmethod
    def forward(ctx, x):
        return torch.round(x)

    @staticmethod
    def backward(ctx, output_grad):
        return output_grad


class FloorStraightThrough(Function):
    """floor forward, identity backward (reference quantizers.py:22-29)."""

    @staticmethod
    def forward(ctx, x):
        return torch.floor(x)

    @staticmethod
    def backward(ctx, output_grad):
        return output_grad


round_ste_func = RoundStraightThrough.apply
floor_ste_func = FloorStraightThrough.apply


class QuantizerNotInitializedError(Exception):
    """Raised when a quantizer has not initialized"""

    def __init__(self):
        super().__init__('Quantizer has not been initialized yet')


def param_layout(x, n_param_elems, axis, per_channel, param_shape=None):
    """(n_params, inner) of include/tq_hip.h for a tensor `x` and a parameter vector."""
    if n_param_elems == 1:
        return 1, 1
    shape = list(x.shape)
    if axis is not None:
        ax = axis
    elif per_channel:
        ax = 0
    else:
        # plain broadcasting of a parameter tensor against x (torch semantics)
        ps = list(param_shape) if param_shape is not None else [n_param_elems]
        ps = [1] * (len(shape) - len(ps)) + ps
        big = [i for i, s in enumerate(ps) if s != 1]
        if len(big) != 1:
            raise ValueError(f'cannot broadcast quantizer parameters {param_shape} over {shape}')
        ax = big[0]
    if shape[ax] != n_param_elems:
        raise RuntimeError(
            f'quantizer holds {n_param_elems} ranges but dim {ax} of the input has size {shape[ax]}')
    inner = 1
    for s in shape[ax + 1:]:
        inner *= s
    return n_param_elems, inner


class _FakeQuantSTE(Function):
    """y = Q(x) through tq_fake_quant_fwd; backward = straight-through estimator kernel."""

    @staticmethod
    def forward(ctx, x, delta, zero_float, quantizer, n_params, inner):
        be = _hip.backend()
        signed = getattr(quantizer, '_signed', None)
        y, _ = be.fake_quant(x, delta, zero_float, signed, quantizer.n_bits,
                             quantizer.symmetric, quantizer.scale_domain == 'log', quantizer.eps,
                             n_params, inner)
        ctx.save_for_backward(x, delta, zero_float)
        ctx.cfg = (signed, quantizer.n_bits, quantizer.symmetric,
                   quantizer.scale_domain == 'log', quantizer.eps, n_params, inner)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, delta, zero_float = ctx.saved_tensors
        signed, n_bits, symmetric, log_domain, eps, n_params, inner = ctx.cfg
        want_p = ctx.needs_input_grad[1] or (zero_float is not None and ctx.needs_input_grad[2])
        gx, gd, gz = _hip.backend().fake_quant_bwd(
            x, grad_y, delta, zero_float, signed, n_bits, symmetric, log_domain, eps, n_params,
            inner, param_grads=want_p)
        if gd is not None:
            gd = gd.view(delta.shape)
            gz = gz.view(zero_float.shape) if zero_float is not None else None
        return (gx if ctx.needs_input_grad[0] else None,
                gd if ctx.needs_input_grad[1] else None,
                gz if (zero_float is not None and ctx.needs_input_grad[2]) else None,
                None, None, None)


class QuantizerBase(nn.Module):
    """Protocol every quantizer implements (reference quantizers.py:36-78)."""

    # Every rebinding of a range buffer / the bit width bumps `_range_gen`; derived caches (int8 weight indices, stacked
    # QKV operands) key on it.  Keys built from data_ptr() + _version alone can collide: a recalibration binds a FRESH
    # tensor at version 0, and the caching allocator likes to hand back the address that was just freed.
    _RANGE_STATE = ('_delta', '_zero_float', '_signed', 'n_bits')

    def __init__(self, n_bits, per_channel=False, axis=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        object.__setattr__(self, '_range_gen', 0)
        self.n_bits = n_bits
        self.per_channel = per_channel
        self.axis = axis

    def __setattr__(self, name, value):
        if name in self._RANGE_STATE:
            object.__setattr__(self, '_range_gen', getattr(self, '_range_gen', 0) + 1)
        super().__setattr__(name, value)

    def range_state_key(self):
        """Changes whenever the quantization grid of this quantizer may have changed (rebinding or in-place update)."""
        d = self._buffers.get('_delta', None) if '_delta' in self._buffers else getattr(self, '_delta', None)
        try:
            ver = None if d is None else d._version
        except RuntimeError:                 # a range set under torch.inference_mode(): no counter -- rebinding still bumps _range_gen
            ver = -1
        return (self._range_gen, ver, self.n_bits, options.CACHE_EPOCH)

    @property
    def is_initialized(self):
        raise NotImplementedError()

    @property
    def x_max(self):
        raise NotImplementedError()

    @property
    def symmetric(self):
        raise NotImplementedError()

    @property
    def x_min(self):
        raise NotImplementedError()

    def forward(self, x_float):
        raise NotImplementedError()

    def _adjust_params_per_axis(self, x):
        raise NotImplementedError()

    def _adjust_params_per_channel(self, x):
        raise NotImplementedError()

    def set_quant_range(self, x_min, x_max):
        raise NotImplementedError()

    def extra_repr(self):
        return (f'n_bits={self.n_bits}, per_channel={self.per_channel}, axis={self.axis}, '
                f'is_initalized={self.is_initialized}')

    def reset(self):
        self._delta = None


class AsymmetricUniformQuantizer(QuantizerBase):
    """x_int = clamp(round(x / scale) + zero_point, 0, 2^n - 1);  y = scale * (x_int - zero_point).

    Parameters
    ----------
    n_bits : int
    scale_domain : 'linear' | 'log'   (log: the stored ``_delta`` is log(delta))
    per_channel : bool                 one range per index of dim 0
    axis : int or None                 one range per index of `axis` (per-embedding / PEG)
    eps : float                        lower bound of the scale
    """

    def __init__(self, n_bits, scale_domain='linear', per_channel=False, axis=None, eps=1e-8):
        super().__init__(n_bits, per_channel)
        assert scale_domain in ('linear', 'log')
        self.register_buffer('_delta', None)
        self.register_buffer('_zero_float', None)
        self.n_bits = n_bits
        self.scale_domain = scale_domain
        self.per_channel = per_channel
        self.axis = axis
        self.eps = eps

    # ---- state -------------------------------------------------------------------------
    @property
    def delta(self):
        if self._delta is None:
            raise QuantizerNotInitializedError()
        return self._delta

    @property
    def zero_float(self):
        if self._zero_float is None:
            raise QuantizerNotInitializedError()
        return self._zero_float

    @property
    def is_initialized(self):
        return self._delta is not None

    @property
    def symmetric(self):
        return False

    # ---- derived quantities (off the hot path; the kernels recompute them in registers) --
    @property
    def int_min(self):
        return 0.0

    @property
    def int_max(self):
        return 2.0 ** self.n_bits - 1

    @property
    def scale(self):
        if self.scale_domain == 'log':
            return torch.exp(self.delta)
        return torch.clamp(self.delta, min=self.eps)

    @property
    def zero_point(self):
        return torch.clamp(round_ste_func(self.zero_float), self.int_min, self.int_max)

    @property
    def x_max(self):
        return self.scale * (self.int_max - self.zero_point)

    @property
    def x_min(self):
        return self.scale * (self.int_min - self.zero_point)

    # ---- kernels -------------------------------------------------------------------------
    def _layout(self, x):
        d = self.delta
        return param_layout(x, d.numel(), self.axis, self.per_channel, d.shape)

    def to_integer_forward(self, x_float):
        """Integer grid indices as an integer-valued float tensor (reference :172-187)."""
        n_params, inner = self._layout(x_float)
        _, idx = _hip.backend().fake_quant(
            x_float, self.delta, self._zero_float, getattr(self, '_signed', None), self.n_bits,
            self.symmetric, self.scale_domain == 'log', self.eps, n_params, inner,
            want_y=False, idx_dtype=torch.float32)
        return idx

    def forward(self, x_float):
        """Quantize-dequantize in one fused kernel (reference :189-211)."""
        if self.axis is not None:
            self._adjust_params_per_axis(x_float)
        if self.per_channel:
            self._adjust_params_per_channel(x_float)
        # state straight from the registries (a fixed-range call is launch-bound: nn.Module.__getattr__ -- and the
        # AttributeError it builds for the `_signed` an asymmetric quantizer does not have -- was a quarter of its
        # host time)
        bufs = self._buffers
        delta = bufs.get('_delta')
        if delta is None:
            delta = self.delta                  # trainable range (nn.Parameter) or not initialised (raises)
        zf = bufs.get('_zero_float')
        if zf is None and not self.symmetric:
            zf = self._zero_float
        n = delta.numel()
        n_params, inner = (1, 1) if n == 1 else param_layout(x_float, n, self.axis, self.per_channel, delta.shape)
        needs_grad = torch.is_grad_enabled() and (
            x_float.requires_grad or delta.requires_grad or (zf is not None and zf.requires_grad))
        if needs_grad:
            return _FakeQuantSTE.apply(x_float, delta, zf, self, n_params, inner)
        y, _ = _hip.backend().fake_quant(
            x_float, delta, zf, bufs.get('_signed'), self.n_bits, self.symmetric,
            self.scale_domain == 'log', self.eps, n_params, inner)
        return y

    def _adjust_params_per_axis(self, x_float):
        # reference :213-217 (a symmetric quantizer has no _zero_float and fails here, as upstream)
        d = self._delta
        if d.dim() == x_float.dim() and d.numel() == d.shape[self.axis] and self._zero_float.shape == d.shape:
            return                       # already in broadcast layout: skip two views + two buffer rebinds per call
        shape = [1] * self.axis + [-1] + [1] * (x_float.dim() - self.axis - 1)
        self._delta = d.view(shape)
        self._zero_float = self._zero_float.view(shape)

    def _adjust_params_per_channel(self, x):
        # reference :219-232
        if x.ndim != self.delta.ndim:
            shape = [-1] + [1] * (x.ndim - 1)
            self._delta = self.delta.view(shape)
            if self._zero_float is not None:
                self._zero_float = self._zero_float.view(shape)

    def _check_range_shape(self, x_min):
        # reference :252-256
        n = x_min.numel() if torch.is_tensor(x_min) else 1
        if torch.is_tensor(x_min) and x_min.dim() > 0 and n > 1 and not self.per_channel \
                and self.axis is None:
            raise ValueError('x_min and x_max must be a float or 1-D Tensor'
                             ' for per-tensor quantization (per_channel=False)')

    def set_quant_range(self, x_min, x_max):
        """(x_min, x_max) -> (_delta, _zero_float) on the device (reference :263-282)."""
        self._check_range_shape(x_min)
        delta, zero_float = _hip.backend().set_range_asym(
            x_min, x_max, self.n_bits, self.eps, self.scale_domain == 'log')
        self._delta = delta.detach()
        self._zero_float = zero_float.detach()

    def make_range_trainable(self):
        if not isinstance(self._delta, nn.Parameter):
            self._delta = nn.Parameter(self._delta)
            self._zero_float = nn.Parameter(self._zero_float)


class SymmetricUniformQuantizer(AsymmetricUniformQuantizer):
    """Zero-point-free variant: signed grid if the range contains negative values
    (reference quantizers.py:291-349)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.register_buffer('_signed', None)

    @property
    def signed(self):
        if self._signed is None:
            raise QuantizerNotInitializedError()
        return self._signed.item()

    @property
    def symmetric(self):
        return True

    @property
    def int_min(self):
        return -(2.0 ** (self.n_bits - 1)) if self.signed else 0

    @property
    def int_max(self):
        return 2.0 ** (self.n_bits - self.signed) - 1

    @property
    def zero_point(self):
        return 0.0

    def set_quant_range(self, x_min, x_max):
        self._check_range_shape(x_min)
        delta, signed = _hip.backend().set_range_sym(
            x_min, x_max, self.n_bits, self.eps, self.scale_domain == 'log')
        self._signed = signed
        self._delta = delta.detach()

    def make_range_trainable(self):
        if not isinstance(self._delta, nn.Parameter):
            self._delta = nn.Parameter(self._delta)


QMethodMap = namedtuple('QMethodMap', ['value', 'cls'])


class QMethods(Enum):
    symmetric_uniform = QMethodMap(0, SymmetricUniformQuantizer)
    asymmetric_uniform = QMethodMap(1, AsymmetricUniformQuantizer)

    @property
    def cls(self):
        return self.value.cls

    @classmethod
    def list(cls):
        return [m.name for m in cls]
