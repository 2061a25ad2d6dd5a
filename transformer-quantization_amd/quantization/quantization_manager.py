"""Per-tensor quantization state machine: one quantizer + one range estimator.

Same surface as the reference's ``quantization/quantization_manager.py`` (``Qstates``,
``QuantizationManager`` with ``state / quantizer / range_estimator / axis / n_groups / n_bits``
and the ``estimate_ranges / fix_ranges / learn_ranges / estimate_ranges_train / reset_ranges /
set_quant_range / forward`` methods, reference :19-112).  In the estimating states a forward is

    estimator(x)  ->  set_quant_range  ->  quantizer(x)

i.e. ``tq_minmax`` (+ finalize) -> ``tq_range_update`` -> ``tq_set_range_*`` -> ``tq_fake_quant_fwd``
on the current HIP stream with no host synchronisation; in ``fix_ranges`` only the last launch runs.
"""
from enum import Enum

import torch
from torch import nn

from quantization import _hip
from quantization import distributed as tq_dist
from quantization import options
from quantization import provenance
from quantization.quantizers import (
    AsymmetricUniformQuantizer,
    QMethods,
    QuantizerNotInitializedError,
    SymmetricUniformQuantizer,
)
from quantization.range_estimators import (
    AllMinMaxEstimator,
    CurrentMinMaxEstimator,
    RangeEstimators,
    RunningMinMaxEstimator,
)

# Estimating forwards of these (estimator, quantizer) pairs run as ONE C call
# (tq_calibrate_minmax: 4 launches) instead of estimator -> set_quant_range -> quantizer
# (3 calls, 5 launches).  Exact types only: subclasses keep the layered path.
FUSED_CALIBRATION = True
_FUSED_ESTIMATORS = {CurrentMinMaxEstimator: _hip.EST_CURRENT, AllMinMaxEstimator: _hip.EST_ALL,
                     RunningMinMaxEstimator: _hip.EST_RUNNING}
_FUSED_QUANTIZERS = (AsymmetricUniformQuantizer, SymmetricUniformQuantizer)
# Fixed-range forwards of per-tensor quantizers skip the generic Python route (QuantizationManager._fixed_fast)
FAST_FIXED_FORWARD = True
_FAST_DTYPES = _hip._DTYPES
from torch.nn.modules import module as _nn_module  # noqa: E402
_GLOBAL_FWD_HOOKS, _GLOBAL_FWD_PRE_HOOKS = _nn_module._global_forward_hooks, _nn_module._global_forward_pre_hooks
_current_device = getattr(torch._C, '_cuda_getDevice', None) or torch.cuda.current_device
_raw_stream = _hip._raw_stream or (lambda dev: torch.cuda.current_stream(dev).cuda_stream)


class Qstates(Enum):
    estimate_ranges = 0        # ranges follow the data in train and eval mode
    fix_ranges = 1             # ranges are frozen
    learn_ranges = 2           # quantizer parameters are nn.Parameters
    estimate_ranges_train = 3  # ranges follow the data in train mode only


class QuantizationManager(nn.Module):
    """Owns the quantizer selected by `qmethod` and the estimator selected by `init`.

    Parameters
    ----------
    qmethod : QMethods member
    init : RangeEstimators member
    per_channel : bool        one range per index of dim 0
    axis, n_groups            per-embedding / per-embedding-group activation quantization
    x_min, x_max              optional fixed range (skips estimation)
    qparams : dict            forwarded to the quantizer (n_bits, scale_domain, ...)
    init_params : dict        forwarded to the estimator (momentum, num_candidates, ...)
    """

    def __init__(self, qmethod=QMethods.symmetric_uniform, init=RangeEstimators.current_minmax,
                 per_channel=False, axis=None, n_groups=None, x_min=None, x_max=None, qparams=None,
                 init_params=None):
        super().__init__()
        object.__setattr__(self, '_fast_plan', None)      # launch plan of the fixed-range fast path (never copied / pickled)
        self.state = Qstates.estimate_ranges
        self.qmethod = qmethod
        self.init = init
        self.per_channel = per_channel
        self.axis = axis
        self.n_groups = n_groups
        self.qparams = qparams if qparams else {}
        self.init_params = init_params if init_params else {}
        self.range_estimator = None

        self.quantizer = self.qmethod.cls(per_channel=per_channel, axis=axis, **self.qparams)

        if x_min is not None and x_max is not None:
            self.set_quant_range(x_min, x_max)
            self.state = Qstates.fix_ranges
        else:
            self.range_estimator = self.init.cls(
                per_channel=self.per_channel, quantizer=self.quantizer, axis=self.axis,
                n_groups=self.n_groups, **self.init_params)

    @property
    def n_bits(self):
        return self.quantizer.n_bits

    def estimate_ranges(self):
        self.state = Qstates.estimate_ranges

    def fix_ranges(self):
        if not self.quantizer.is_initialized:
            raise QuantizerNotInitializedError()
        self.state = Qstates.fix_ranges

    def learn_ranges(self):
        self.quantizer.make_range_trainable()
        self.state = Qstates.learn_ranges

    def estimate_ranges_train(self):
        self.state = Qstates.estimate_ranges_train

    def reset_ranges(self):
        self.range_estimator.reset()
        self.quantizer.reset()
        self.estimate_ranges()

    def _estimating(self):
        return self.state == Qstates.estimate_ranges or (
            self.state == Qstates.estimate_ranges_train and self.training)

    def _fused_estimating_forward(self, x):
        """estimator(x) -> set_quant_range -> quantizer(x) as one backend call (two around the all-reduce when
        calibration is sharded), or None when the configuration needs the layered path (percentiles, custom
        classes, autograd, CPU tensors, > 4096 ranges).  Leaves exactly the state the layered path leaves."""
        mods = self._modules
        est, q = mods.get('range_estimator'), mods['quantizer']
        mode = _FUSED_ESTIMATORS.get(type(est))
        be = _hip.backend()
        sharded = tq_dist.is_enabled()
        if (mode is None or type(q) not in _FUSED_QUANTIZERS or not FUSED_CALIBRATION
                or not hasattr(be, 'calibrate_stats' if sharded else 'calibrate_minmax')
                or not (x.is_cuda or getattr(be, 'accepts_cpu', False)) or x.dtype == torch.float64   # --double: layered
                or getattr(est, 'percentile', None) or '_delta' not in q._buffers     # trainable ranges
                or (torch.is_grad_enabled() and x.requires_grad)):
            return None
        axis = None if mode == _hip.EST_ALL else est.axis
        n_groups = 0
        if axis is not None:
            n_params, inner = x.shape[axis], 1
            for s in x.shape[axis + 1:]:
                inner *= s
            if est.n_groups is not None:
                assert est.n_groups > 0 and n_params % est.n_groups == 0
                n_groups = est.n_groups
        elif est.per_channel:
            n_params, inner = x.shape[0], x.numel() // max(x.shape[0], 1)
        else:
            n_params, inner = 1, 1
        if n_params > be.CALIB_MAX_PARAMS or x.numel() == 0:
            return None
        if n_params > 1 and not q.per_channel and q.axis is None:
            return None                      # layered path raises the reference's ValueError
        order = None
        if n_groups and mode == _hip.EST_CURRENT and est.ranges is not None:
            order = be.argsort(est.ranges)
        ebufs = est._buffers                 # (registered buffers, read through the registry: see forward)
        prev_min = ebufs['current_xmin'] if mode != _hip.EST_CURRENT else None
        prev_max = ebufs['current_xmax'] if mode != _hip.EST_CURRENT else None
        if prev_min is not None and (prev_min.dtype != torch.float32 or prev_max.dtype != torch.float32):
            return None                      # state left by a float64 pass: the layered path promotes like torch
        out = None
        if options.INPLACE_CALIBRATION_STATE:
            out = self._inplace_state(est, q, n_params, x.device)
        box = comm = None
        if sharded:
            box = tq_dist.mailbox_for(2 * n_params) if hasattr(be, 'calibrate_minmax_mailbox') else None
            if box is None and hasattr(be, 'calibrate_minmax_rccl'):
                comm = tq_dist.raw_comm_for(x)
        if box is not None:
            # statistics -> P2P mailbox all-reduce -> update + quantize as one C call (3-4 launches, no host work between)
            cur_min, cur_max, delta, zero_float, signed, y = be.calibrate_minmax_mailbox(
                box, x, n_params, inner, mode, prev_min, prev_max, getattr(est, 'momentum', 0.0), n_groups, order,
                q.n_bits, q.symmetric, q.eps, q.scale_domain == 'log', out=out)
            tq_dist.count_mailbox_exchange(8 * n_params)
        elif comm is not None:
            # statistics -> ncclAllReduce(MAX) on the raw communicator -> update + quantize as one C call: no c10d
            cur_min, cur_max, delta, zero_float, signed, y = be.calibrate_minmax_rccl(
                comm, x, n_params, inner, mode, prev_min, prev_max, getattr(est, 'momentum', 0.0),
                n_groups, order, q.n_bits, q.symmetric, q.eps, q.scale_domain == 'log', out=out)
            tq_dist.count_raw_exchange(8 * n_params)
        elif sharded:
            # split at the exchange: local [-min | max] -> one in-place MAX all-reduce -> update + quantize
            stats = tq_dist.sync_max_inplace(be.calibrate_stats(x, n_params, inner))
            cur_min, cur_max, delta, zero_float, signed, y = be.calibrate_apply(
                stats, x, n_params, inner, mode, prev_min, prev_max, getattr(est, 'momentum', 0.0), n_groups,
                order, q.n_bits, q.symmetric, q.eps, q.scale_domain == 'log', out=out)
        else:
            cur_min, cur_max, delta, zero_float, signed, y = be.calibrate_minmax(
                x, n_params, inner, mode, prev_min, prev_max, getattr(est, 'momentum', 0.0), n_groups, order,
                q.n_bits, q.symmetric, q.eps, q.scale_domain == 'log', out=out)
        # registered buffers: rebinding through the dict skips nn.Module.__setattr__'s type dispatch
        # (4 rebinds per call x 161 quantizers per calibration batch)
        ebufs['current_xmin'], ebufs['current_xmax'] = cur_min, cur_max
        object.__setattr__(q, '_range_gen', q._range_gen + 1)          # the dict rebinds below bypass __setattr__
        q._buffers['_delta'] = delta
        if q.symmetric:
            q._buffers['_signed'] = signed
        else:
            q._buffers['_zero_float'] = zero_float
        # same buffer shapes as quantizer.forward leaves behind ([1,1,d] / [C,1,..] views)
        if q.axis is not None:
            q._adjust_params_per_axis(x)
        if q.per_channel:
            q._adjust_params_per_channel(x)
        return y

    @staticmethod
    def _inplace_state(est, q, n_params, device):
        """Existing (state, parameter) buffers to overwrite, or None while any of them is missing or has
        another size (first batch, changed layout): the step then allocates fresh ones."""
        bufs = [est.current_xmin, est.current_xmax, q._delta,
                None if q.symmetric else q._zero_float, q._signed if q.symmetric else None]
        need = [True, True, True, not q.symmetric, q.symmetric]
        for t, wanted in zip(bufs, need):
            if not wanted:
                continue
            if (t is None or t.device != device or not t.is_contiguous()
                    or t.numel() != (1 if t.dtype == torch.bool else n_params)
                    or t.dtype not in (torch.float32, torch.bool)):
                return None
        return tuple(bufs)

    def quantize(self, x):
        """`self(x)` for callers inside the package (QuantizedActivation): identical result, but a fixed-range call of a
        hook-free manager goes straight to the launch plan (`_fixed_fast`) without a second nn.Module.__call__ --
        ~0.6 us of the ~7 us a launch-bound quantizer call costs on the host.  Anything out of the ordinary (hooks on
        this manager, global module hooks, the integer path, PEG range collection, an ineligible quantizer) takes
        `self(x)`."""
        if (self.state is Qstates.fix_ranges and FAST_FIXED_FORWARD
                and not (self._forward_hooks or self._forward_pre_hooks or _GLOBAL_FWD_HOOKS or _GLOBAL_FWD_PRE_HOOKS)):
            mods = self._modules
            est = mods.get('range_estimator')
            if est is None or not est.per_group_range_estimation:
                q = mods['quantizer']
                if x.dtype is not torch.float32 or not options.int8_active():
                    y = self._fixed_fast(x, q)
                elif type(q) is AsymmetricUniformQuantizer and q.n_bits <= 8 and q.scale_domain == 'linear':
                    y = self._fixed_fast(x, q, with_idx=True)      # integer route: y and its int8 indices, tagged
                else:
                    y = None
                if y is not None:
                    return y
        return self(x)

    def forward(self, x):
        # (sub-modules through the registry: nn.Module.__getattr__ is the slow path of attribute access, and a
        # fixed-range call is launch-bound)
        mods = self._modules
        est = mods.get('range_estimator')
        if est is not None and est.per_group_range_estimation:
            est(x)          # PEG phase 1: only collect per-dimension ranges, pass x through
            return x
        if self.state == Qstates.fix_ranges:
            return self._fixed_forward(x, mods['quantizer'])
        if self._estimating():
            if est is None:
                raise RuntimeError('this manager was built with a fixed range: no estimator to run')
            y = self._fused_estimating_forward(x)
            if y is None:
                cur_xmin, cur_xmax = est(x)
                self.set_quant_range(cur_xmin, cur_xmax)
                y = mods['quantizer'](x)
            if options.INT8_CALIBRATION and options.int8_active():
                # calibrating forward on the integer route: what this call returns lies on the grid it has just set -- the
                # consuming integer Linear derives the int8 indices itself (one index-only launch: no estimator kernel
                # emits them) and reads the parameters from the device buffers recorded here
                q = mods['quantizer']
                if (type(q) is AsymmetricUniformQuantizer and q.n_bits <= 8 and q.scale_domain == 'linear'
                        and y.dtype is torch.float32 and q._delta is not None and q._delta.numel() == 1):
                    provenance.tag(y, q)
            return y
        return mods['quantizer'](x)

    def __getstate__(self):
        # the launch plan holds ctypes references into this process's descriptor cache: rebuilt on demand, never copied
        state = self.__dict__.copy()
        state['_fast_plan'] = None
        return state

    def _fixed_fast(self, x, q, with_idx=False):
        """Fixed per-tensor range, plain ROCm tensor, no autograd: torch.empty_like + one foreign call.  Bit-identical to
        q(x) -- it IS the same entry point with the same descriptor, minus the Python of the generic route (module
        dispatch, argument marshalling, layout checks) per call.  Returns None whenever anything is out of the ordinary
        (hooks on the quantizer, per-axis / per-channel ranges, trainable or missing ranges, range buffers on another
        device than x, another device than the current one, non-contiguous input, a backend double): the caller then
        takes the generic route.

        The launch plan is revalidated per call by IDENTITY of the range buffers it was built for (the plan holds them, so
        an address cannot be recycled under it) plus the quantizer's rebinding counter: `.to()` / `load_state_dict` /
        a new calibration all produce either new tensor objects or a new `_range_gen`; in-place value updates need no
        new plan (the kernel reads the values from the buffers)."""
        bufs = q._buffers
        plan = self._fast_plan
        try:
            stale = (plan is None or plan[0] != q._range_gen or bufs.get('_delta') is not plan[1]
                     or bufs.get('_zero_float') is not plan[2] or bufs.get('_signed') is not plan[3] or q.eps != plan[4]
                     or q.scale_domain != plan[5] or _hip._backend is not plan[6] or type(q) is not plan[7])
        except AttributeError:          # a quantizer class of the user's without eps / scale_domain: never eligible
            stale = True
        if stale:
            plan = self._make_fast_plan(q)
        call = plan[9]
        if call is None:
            return None
        dev = plan[8]
        # (get_device() is -1 for a host tensor: one call covers `is_cuda` and the device index)
        if (x.get_device() != dev or x.dtype not in _FAST_DTYPES or q._forward_hooks or q._forward_pre_hooks
                or (x.requires_grad and torch.is_grad_enabled()) or _current_device() != dev or not x.is_contiguous()):
            return None
        y = torch.empty_like(x)
        ref = plan[10]
        if with_idx:
            # the integer route's producers: the same launch also writes int8(index - 128) for the consuming integer GEMM
            # (fp32 tensors on <= 8-bit asymmetric linear-domain grids only: the caller checked)
            idx = torch.empty(x.shape, dtype=torch.int8, device=x.device)
            ip, it = idx.data_ptr(), _hip.IDX_I8_M128
        else:
            idx, ip, it = None, 0, 0
        if type(ref) is tuple:          # CPython stub (csrc_py/tq_fastcall.c): (entry address, descriptor address)
            rc = call(ref[0], x.data_ptr(), y.data_ptr(), ip, it, x.numel(), _FAST_DTYPES[x.dtype], ref[1], _raw_stream(dev))
        else:                           # ctypes: the same entry point, marshalled
            rc = call(x.data_ptr(), y.data_ptr(), ip or None, it, x.numel(), _FAST_DTYPES[x.dtype], ref, _raw_stream(dev))
        if rc != 0:
            _hip._check(rc, plan[12])
        if with_idx:
            provenance.tag(y, q, idx)
        return y

    def _make_fast_plan(self, q):
        """(generation, delta, zero_float, signed, eps, scale_domain, backend, quantizer type, device index, call,
        descriptor reference, descriptor owner, library) -- `call` is None when this quantizer is not eligible (the
        verdict is cached under the same validity conditions as a plan)."""
        bufs = q._buffers
        delta, zf, sg = bufs.get('_delta'), bufs.get('_zero_float'), bufs.get('_signed')
        be = _hip.backend()
        head = (q._range_gen, delta, zf, sg, getattr(q, 'eps', None), getattr(q, 'scale_domain', None), be, type(q))
        ok = (delta is not None and delta.is_cuda and type(q) in _FUSED_QUANTIZERS and delta.numel() == 1
              and q.axis is None and not q.per_channel and be is not None and hasattr(be, 'fixed_quant_plan')
              and not (q.symmetric and sg is None) and not (not q.symmetric and zf is None) and not delta.requires_grad
              and (zf is None or zf.device == delta.device) and (sg is None or sg.device == delta.device))
        if ok:
            call, ref, owner = be.fixed_quant_plan(delta, zf, sg, q.n_bits, q.symmetric, q.scale_domain == 'log', q.eps)
            plan = head + (delta.device.index, call, ref, owner, be.lib)
        else:
            plan = head + (-2, None, None, None, None)
        object.__setattr__(self, '_fast_plan', plan)
        return plan

    def _fixed_forward(self, x, q):
        if not options.int8_active():
            y = self._fixed_fast(x, q) if FAST_FIXED_FORWARD else None
            return q(x) if y is None else y
        y = None
        wants_idx = (type(q) is AsymmetricUniformQuantizer and q.n_bits <= 8 and x.dtype is torch.float32
                     and q.scale_domain == 'linear')
        if wants_idx and FAST_FIXED_FORWARD:
            y = self._fixed_fast(x, q, with_idx=True)      # launch plan of the plain fixed-range call + the index pointer
            if y is not None:
                return y                                   # (tagged with its indices inside)
        if wants_idx:
            y = self._fixed_forward_with_indices(x)
        if y is None:
            # no index output for this tensor (bf16 / fp16 storage, > 8 bits, per-axis ranges, ...): the plain launch
            y = self._fixed_fast(x, q) if FAST_FIXED_FORWARD else None
            if y is None:
                y = q(x)
        # provenance record: lets a consumer (the fused integer Linear, also under autograd in QAT) recover the
        # exact grid indices of this tensor from the quantizer that produced it (quantization/provenance.py);
        # only the integer fast paths consume it
        provenance.tag(y, q, provenance.indices_of(y))
        return y

    def _fixed_forward_with_indices(self, x):
        """Fixed per-tensor asymmetric <= 8-bit quantizer feeding integer Linears: one launch writes
        the dequantised tensor AND its int8 indices (minus 128), saving the consumer's re-quantisation."""
        q = self.quantizer
        if (type(q) is not AsymmetricUniformQuantizer or q.n_bits > 8 or q._delta.numel() != 1
                or q.scale_domain != 'linear' or not _hip.on_device(x) or x.dtype != torch.float32
                or (torch.is_grad_enabled() and x.requires_grad)):
            return None
        be = _hip.backend()
        if not hasattr(be, 'fake_quant_int8'):
            return None
        y, idx = be.fake_quant_int8(x, q._delta, q._zero_float, q.n_bits, q.eps)
        return provenance.tag(y, q, idx)

    def set_quant_range(self, x_min, x_max):
        self.quantizer.set_quant_range(x_min, x_max)

    def extra_repr(self):
        return 'state={}'.format(self.state.name)
