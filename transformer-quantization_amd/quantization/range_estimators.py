"""Online range estimators on top of the gfx950 statistics / search kernels.

Drop-in for the reference's ``quantization/range_estimators.py`` (same classes, constructor
arguments, ``current_xmin`` / ``current_xmax`` buffers, ``per_group_range_estimation`` /
``ranges`` attributes and exceptions).  What runs where:

* batch min/max (per tensor, per channel, per embedding dim)   -> ``tq_minmax``  (K4/K5)
  reference range_estimators.py:82-85, 114-116, 118-130, 142-143, 153-160, 178-207
* group fold / range-sorted permutation / EMA / all-time merge  -> ``tq_range_update``
  reference :87-112, :162-167, :183-193, :209-214 (the reference's dense permutation matmul
  ``P.mm(x)`` is never formed: only the [d] statistics are permuted)
* PEG phase-1 ``ranges``                                          -> ``tq_axis_ranges`` (:68-80)
* MSE / cross-entropy grid search: every candidate quantizer is evaluated in ONE pass over the
  tensor (``tq_mse_candidates`` / ``tq_xent_candidates``), the argmin and threshold lookup stay
  on the device (``tq_argmin_select``).  reference :248-256, :287-294, :356-420, :498-502.
  The candidate (scale, zero_point, int_min, int_max) table is O(num_candidates) scalar work and
  is prepared on the host in numpy fp32 with the reference's operation order.
* golden-section search keeps ``scipy.optimize.minimize_scalar`` as the owner of the iterate
  sequence (reference :321, :429, :449, :458); each loss evaluation is one kernel launch.  K INDEPENDENT searches
  (the 102 weight tensors of a BERT-base under the README recipe) advance in lock step (`golden_section_lockstep`):
  scipy's bounded Brent restated as a resumable generator (pinned against scipy), one queue of launches + ONE
  device->host copy per round instead of per evaluation.

When ``quantization.distributed`` is enabled, the per-rank statistics (min/max, candidate
losses) are all-reduced over RCCL before the state update, so every rank ends up with the
ranges of the concatenated batch.
"""
import math
import weakref
from collections import namedtuple
from enum import Enum

import numpy as np
import torch
from scipy.optimize import minimize_scalar
from torch import nn

from quantization import _hip
from quantization import distributed as tq_dist


def _hip_version(t):
    try:
        return t._version
    except RuntimeError:
        return None


class NoDataPassedError(Exception):
    """Raised data has been passed inot the Range Estimator."""

    def __init__(self):
        super().__init__('Data must be pass through the range estimator to be initialized')


class RangeEstimatorBase(nn.Module):
    def __init__(self, per_channel=False, quantizer=None, axis=None, n_groups=None, *args,
                 **kwargs):
        super().__init__(*args, **kwargs)
        self.register_buffer('current_xmin', None)
        self.register_buffer('current_xmax', None)
        self.per_channel = per_channel
        self.quantizer = quantizer
        self.axis = axis
        self.n_groups = n_groups

        self.per_group_range_estimation = False
        self.ranges = None

    def forward(self, x):
        """Update and return (current_xmin, current_xmax)."""
        raise NotImplementedError()

    def reset(self):
        self.current_xmin = None
        self.current_xmax = None

    def __repr__(self):
        # hide the shared quantizer sub-module, like the reference (:53-59)
        lines = self.extra_repr().split('\n')
        extra = lines[0] if len(lines) == 1 else '\n  ' + '\n  '.join(lines) + '\n'
        return self._get_name() + '(' + extra + ')'

    # ---- shared kernel plumbing --------------------------------------------------------
    @staticmethod
    def _local_minmax(x, n_params, inner):
        """Batch statistics of this rank's shard.  A rank whose shard is EMPTY (calibration batch smaller than the
        world size, e.g. the README recipe's --est-ranges-batch-size 1 over 8 GPUs) contributes the identity of the
        MAX all-reduce of [-min | max]; without sharding an empty tensor is an error, as in the reference."""
        if x.numel() == 0 and tq_dist.is_enabled():
            dt = torch.float64 if x.dtype == torch.float64 else torch.float32
            shape = () if n_params == 1 else (n_params,)
            return (torch.full(shape, float('inf'), dtype=dt, device=x.device),
                    torch.full(shape, float('-inf'), dtype=dt, device=x.device))
        return _hip.backend().minmax(x, n_params, inner)

    def _axis_stats(self, x):
        """min/max per index of self.axis, all-reduced across ranks if sharded."""
        inner = 1
        for s in x.shape[self.axis + 1:]:
            inner *= s
        mn, mx = self._local_minmax(x, x.shape[self.axis], inner)
        return tq_dist.sync_minmax(mn, mx)

    def _channel_stats(self, x):
        mn, mx = self._local_minmax(x, x.shape[0], x.numel() // max(x.shape[0], 1))
        return tq_dist.sync_minmax(mn, mx)

    def _tensor_stats(self, x):
        mn, mx = self._local_minmax(x, 1, 1)
        return tq_dist.sync_minmax(mn, mx)


class CurrentMinMaxEstimator(RangeEstimatorBase):
    """Range of the current batch only (reference :62-145)."""

    def __init__(self, percentile=None, *args, **kwargs):
        self.percentile = percentile
        super().__init__(*args, **kwargs)

    def forward(self, x):
        be = _hip.backend()
        if self.per_group_range_estimation:
            # PEG phase 1: remember max-min per embedding dimension, quantize nothing
            assert self.axis != 0
            mn, mx = self._axis_stats(x)
            self.ranges = be.axis_ranges(mn, mx, first=self.ranges is None)
            return

        if self.axis is not None:
            mn, mx = self._axis_stats(x)
            if self.n_groups is not None:
                ng = self.n_groups
                assert ng > 0 and mn.numel() % ng == 0
                order = be.argsort(self.ranges) if self.ranges is not None else None
                mn, mx = be.range_update(_hip.EST_CURRENT, mn, mx, None, None, n_groups=ng,
                                         order=order)
            self.current_xmin, self.current_xmax = mn.detach(), mx.detach()

        elif self.per_channel:
            if self.percentile:
                self.current_xmin, self.current_xmax = _percentile_rows(
                    x.view(x.shape[0], -1), (self.percentile, 100 - self.percentile))
            else:
                mn, mx = self._channel_stats(x)
                self.current_xmin, self.current_xmax = mn.detach(), mx.detach()

        else:
            if self.percentile:
                # NB asymmetric on purpose: the reference uses (p, 100) here (:136, quirk q6)
                lo, hi = _percentile_rows(x.reshape(1, -1), (self.percentile, 100))
                self.current_xmin, self.current_xmax = lo, hi
            else:
                mn, mx = self._tensor_stats(x)
                self.current_xmin, self.current_xmax = mn.detach(), mx.detach()

        return self.current_xmin, self.current_xmax


def _percentile_rows(rows, q):
    """np.percentile(rows, q, axis=-1) (method 'linear') without sorting and without leaving the device.

    The reference moves the whole tensor to the host and calls numpy (:121-140).  numpy's result depends on exactly two
    order statistics per percentile -- the elements at floor and ceil of the virtual index (n - 1) p / 100 -- which
    `tq_order_stats` selects exactly (radix select, csrc/tq_select.hip); they are then combined with numpy's own
    arithmetic: the difference b - a in the data dtype (fp32), the interpolation in float64, the result narrowed to
    fp32 by ``torch.Tensor(...)`` -- bit-identical to numpy on the same values.  (A -0.0 / +0.0 tie may come out with
    the other sign than numpy's sort would leave at that position; the values compare equal.)"""
    rows = rows.detach()
    if rows.dtype == torch.float64:
        rows = rows.float()          # --double: the selection kernel is fp32 (the round-3 sort path cast the same way)
    n = rows.shape[-1]
    plan = []
    for p in q:
        vidx = (n - 1) * (p / 100.0)
        lo_i = int(math.floor(vidx))
        plan.append((lo_i, min(lo_i + 1, n - 1), vidx - lo_i))
    stats = _hip.backend().order_stats(rows.reshape(-1, n), [i for lo_i, hi_i, _ in plan for i in (lo_i, hi_i)])
    out = []
    for k, (_, _, t) in enumerate(plan):
        a, b = stats[:, 2 * k], stats[:, 2 * k + 1]
        diff = (b - a).double()
        val = a.double() + diff * t if t < 0.5 else b.double() - diff * (1 - t)
        out.append(val.float())
    return out[0], out[1]


class AllMinMaxEstimator(RangeEstimatorBase):
    """All-time min/max over the batches seen; ignores axis / n_groups like the reference
    (:148-169, quirk q5)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)

    def forward(self, x):
        mn, mx = self._channel_stats(x) if self.per_channel else self._tensor_stats(x)
        self.current_xmin, self.current_xmax = _hip.backend().range_update(
            _hip.EST_ALL, mn, mx, self.current_xmin, self.current_xmax)
        return self.current_xmin, self.current_xmax


class RunningMinMaxEstimator(RangeEstimatorBase):
    """Exponential moving average of the batch ranges (reference :172-216)."""

    def __init__(self, momentum=0.9, *args, **kwargs):
        self.momentum = momentum
        super().__init__(*args, **kwargs)

    def forward(self, x):
        n_groups = 0
        if self.axis is not None:
            mn, mx = self._axis_stats(x)
            if self.n_groups is not None:
                assert self.n_groups > 0 and mn.numel() % self.n_groups == 0
                n_groups = self.n_groups
        elif self.per_channel:
            mn, mx = self._channel_stats(x)
        else:
            mn, mx = self._tensor_stats(x)
        self.current_xmin, self.current_xmax = _hip.backend().range_update(
            _hip.EST_RUNNING, mn, mx, self.current_xmin, self.current_xmax,
            momentum=self.momentum, n_groups=n_groups)
        return self.current_xmin, self.current_xmax


class OptMethod(Enum):
    grid = 1
    golden_section = 2

    @classmethod
    def list(cls):
        return [m.name for m in cls]


# --------------------------------------------------------------------------------------
# candidate tables (host, numpy fp32, reference operation order)
# --------------------------------------------------------------------------------------
def candidate_params(neg_thr, pos_thr, n_bits, symmetric, eps=1e-8, log_domain=False):
    """(scale, zero_point, int_min, int_max) of the temporary quantizer the reference builds for
    a pair of thresholds: set_quant_range (quantizers.py:234-282 / :334-344) followed by the
    scale / zero_point properties (:142-153).  Vectorised over candidates.  log_domain: the reference stores
    log(delta) and quantizes with exp(log(delta)) (no eps clamp, :143-147, :279-282) -- evaluated here with the same
    fp32 torch-CPU log / exp the reference's host path runs, so the table holds the scale its candidates see."""
    f32 = np.float32
    x_min = np.minimum(np.asarray(neg_thr, dtype=np.float64).astype(f32), f32(0.0))
    x_max = np.maximum(np.asarray(pos_thr, dtype=np.float64).astype(f32), f32(eps))
    if symmetric:
        signed = x_min < 0
        int_max = np.where(signed, 2.0 ** (n_bits - 1) - 1, 2.0 ** n_bits - 1).astype(f32)
        int_min = np.where(signed, -(2.0 ** (n_bits - 1)), 0.0).astype(f32)
        delta = (np.maximum(np.abs(x_min), x_max) / int_max).astype(f32)
        zp = np.zeros_like(delta)
    else:
        int_max = np.full_like(x_min, 2.0 ** n_bits - 1)
        int_min = np.zeros_like(x_min)
        delta = ((x_max - x_min) / int_max).astype(f32)
        zero_float = (-x_min / delta).astype(f32)
        zp = np.clip(np.rint(zero_float), f32(0.0), int_max).astype(f32)
    if log_domain:
        # one 0-d tensor per candidate, as upstream: ATen's vectorised and scalar log / exp may differ in the last bit
        flat = np.ascontiguousarray(delta, dtype=f32).reshape(-1)
        scale = np.array([float(torch.exp(torch.log(torch.tensor(v)))) for v in flat], dtype=f32).reshape(np.shape(delta))
    else:
        scale = np.maximum(delta, f32(eps)).astype(f32)
    return np.stack([scale, zp, int_min, int_max], axis=-1).astype(f32)


class MSE_Estimator(RangeEstimatorBase):
    """Clipping thresholds that minimise the quantisation MSE (reference :228-490)."""

    def __init__(self, num_candidates=100, opt_method=OptMethod.grid, range_margin=0.5, *args,
                 **kwargs):
        super().__init__(*args, **kwargs)
        assert opt_method in OptMethod

        self.opt_method = opt_method
        self.num_candidates = num_candidates
        self.max_pos_thr = None
        self.max_neg_thr = None
        self.max_search_range = None
        self.one_sided_dist = None
        self.range_margin = range_margin
        if self.quantizer is None:
            raise NotImplementedError(
                'A Quantizer must be given as an argument to the MSE Range' 'Estimator')
        self.max_int_skew = (2 ** self.quantizer.n_bits) // 4  # for asymmetric quantization

        # Extension, off by default (the reference ignores axis / n_groups here, quirk q5): search one
        # range per embedding group, like MSE_Estimator(per_channel=True) applied to the
        # [n_groups, -1] view of the tensor.  Enable per instance (`est.per_group_search = True`).
        self.per_group_search = False

        self._loss_dev = None      # fp64 [groups, n_cand] on the device
        self._cand_dev = None      # fp32 [n_cand, 4]
        self._thr_dev = None       # fp32 [2, n_cand]
        self._cand_shape = None
        self._memo = None          # (input key, thresholds) left by golden_section_lockstep

    # ---- reference-visible state ---------------------------------------------------------
    @property
    def loss_array(self):
        """numpy view in the reference's layout ([groups, C+1] or [groups, C+1, skew, 2]) with the
        excluded index 0 set to inf.  Host copy -- diagnostics only."""
        if self._loss_dev is None:
            return None
        host = self._loss_dev.cpu().numpy()
        groups = host.shape[0]
        if len(self._cand_shape) == 1:
            full = np.full((groups, self.num_candidates + 1), np.inf)
            full[:, 1:] = host
        else:
            full = np.full((groups, self.num_candidates + 1, self.max_int_skew, 2), np.inf)
            full[:, 1:] = host.reshape((groups,) + self._cand_shape)
        return full

    @property
    def step_size(self):
        if self.one_sided_dist is None:
            raise NoDataPassedError()
        return self.max_search_range / self.num_candidates

    @property
    def _one_dimensional(self):
        return bool(self.one_sided_dist or self.quantizer.symmetric)

    @property
    def optimization_method(self):
        if self.one_sided_dist is None:
            raise NoDataPassedError()
        if self.opt_method == OptMethod.grid:
            return self._perform_1D_search if self._one_dimensional else self._perform_2D_search
        if self.opt_method == OptMethod.golden_section:
            return (self._golden_section_symmetric if self._one_dimensional
                    else self._golden_section_asymmetric)
        raise NotImplementedError('Optimization Method not Implemented')

    # ---- loss evaluation on the device -------------------------------------------------------
    def _grouped(self, data):
        if not (self.per_group_search and self.axis is not None and self.n_groups):
            return False
        if self.axis != data.dim() - 1:
            raise NotImplementedError('per-group MSE search needs the grouped axis to be the last one')
        assert data.shape[-1] % self.n_groups == 0
        return True

    def _rows(self, data):
        if self._grouped(data):
            return self.n_groups
        return len(data) if self.per_channel else 1

    def _batch_losses(self, data, cand_dev, rows, per_row=None):
        """fp64 [rows, n_cand]: this batch's loss per candidate (summed over ranks if sharded)."""
        be = _hip.backend()
        loss = be.zeros_f64((rows, cand_dev.shape[0]), data.device)
        self._launch_loss(be, data, rows, cand_dev, loss,
                          per_row=self.per_channel if per_row is None else per_row)
        return tq_dist.sync_sum(loss)

    def _launch_loss(self, be, data, rows, cand_dev, loss, per_row=False):
        if self._grouped(data) and rows == self.n_groups:
            # extension without a reference order: fp64-accumulated sums
            be.mse_candidates_grouped(data, self.n_groups, cand_dev, loss)
        else:
            # the fp32 value of the reference's two torch.sum calls (:250-256), in ATen's CPU order
            be.mse_candidates_ordered(data, cand_dev, loss, per_row=per_row)

    def _cand_table(self, neg_thr, pos_thr):
        q = self.quantizer
        return candidate_params(neg_thr, pos_thr, q.n_bits, q.symmetric, q.eps, log_domain=q.scale_domain == 'log')

    def loss_fx(self, data, neg_thr, pos_thr, per_channel_loss=False):
        """Loss of ONE candidate as a host value in the precision of the reference's sums (fp32 for fp32 / low-precision
        data, fp64 under --double; golden-section path; reference :248-256): the same
        bits as the reference's `torch.sum(torch.sum(err.view(len(data), -1), dim=1))` on the CPU, so that
        scipy's iterates -- hence the returned thresholds -- follow the reference's."""
        len(data)       # reference :250 views the error as [len(data), -1]: a 0-d slice (per-channel golden section on a
        #                 1-D weight such as LayerNorm's) raises TypeError there, and therefore here
        if not (neg_thr or pos_thr):
            # quirk q7 (reference :292): both thresholds falsy -> the quantizer's current range
            neg_thr, pos_thr = float(self.quantizer.x_min), float(self.quantizer.x_max)
        be = _hip.backend()
        cand = be.candidate_table(self._cand_table([neg_thr], [pos_thr]), data.device)
        rows = len(data) if per_channel_loss else 1
        loss = self._batch_losses(data, cand, rows, per_row=per_channel_loss)
        host = loss.cpu().numpy()
        if data.dtype != torch.float64:
            host = host.astype(np.float32)               # exact: the fp64 cell holds one fp32 value
        # (--double: the reference's sums are float64 themselves, scipy sees the float64 value)
        return host[:, 0] if per_channel_loss else host[0, 0]

    def quantize(self, x_float, x_min=None, x_max=None):
        """Fake-quantize with a temporary per-tensor copy of the quantizer (reference :287-294)."""
        import copy
        temp_q = copy.deepcopy(self.quantizer)
        temp_q.per_channel = False
        if x_min or x_max:
            temp_q.set_quant_range(x_min, x_max)
        return temp_q(x_float)

    # ---- search space ----------------------------------------------------------------------
    def _define_search_range(self, data, stats=None):
        be = _hip.backend()
        self.channel_groups = self._rows(data)
        mn, mx = stats if stats is not None else self._tensor_stats(data)
        data_min, data_max = float(mn), float(mx)            # one host sync, first batch only
        if self._one_dimensional:
            self.max_pos_thr = max(abs(data_min), data_max) + self.range_margin
            self.max_neg_thr = -self.max_pos_thr
            self.max_search_range = self.max_pos_thr
        else:
            self.max_pos_thr = data_max + self.range_margin
            self.max_neg_thr = data_min - self.range_margin
            self.max_search_range = max(abs(self.max_pos_thr), abs(self.max_neg_thr))

        if self.opt_method != OptMethod.grid:
            self._loss_dev = be.zeros_f64((self.channel_groups, 1), data.device)
            self._cand_shape = (1,)
            return

        C = self.num_candidates
        step = self.step_size
        cidx = np.arange(1, C + 1, dtype=np.float64)
        if self._one_dimensional:
            pos = step * cidx                                 # reference :363-364
            neg = np.zeros_like(pos) if self.one_sided_dist else -step * cidx
            self._cand_shape = (C,)
        else:
            # reference :389-399: symmetric interval, integer skew, clipped to the data range
            levels = 2 ** self.quantizer.n_bits - 1
            start = (-step * cidx)[:, None, None]
            finish = (step * cidx)[:, None, None]
            delta = (finish - start) / levels
            shift = np.arange(self.max_int_skew, dtype=np.float64)[None, :, None]
            sign = np.array([1.0, -1.0])[None, None, :]
            skew = sign * shift * delta
            neg = np.maximum(start + skew, self.max_neg_thr).reshape(-1)
            pos = np.minimum(finish + skew, self.max_pos_thr).reshape(-1)
            self._cand_shape = (C, self.max_int_skew, 2)
        self._cand_dev = be.candidate_table(self._cand_table(neg, pos), data.device)
        thr = np.stack([neg.astype(np.float32), pos.astype(np.float32)])
        self._thr_dev = be.candidate_table(thr, data.device)
        self._loss_dev = be.zeros_f64((self.channel_groups, self._cand_dev.shape[0]), data.device)

    # ---- grid searches: one kernel pass for all candidates ---------------------------------
    def _grid_search(self, data):
        be = _hip.backend()
        self._loss_dev = self._loss_dev + self._batch_losses(data, self._cand_dev,
                                                             self.channel_groups)
        xmin, xmax, _ = be.argmin_select(self._loss_dev, self._thr_dev[0], self._thr_dev[1])
        if self._grouped(data):
            gs = data.shape[-1] // self.n_groups
            xmin, xmax = xmin.repeat_interleave(gs), xmax.repeat_interleave(gs)
        self.current_xmin, self.current_xmax = xmin, xmax

    def _perform_1D_search(self, data):
        self._grid_search(data)

    def _perform_2D_search(self, data):
        self._grid_search(data)

    # ---- golden section: scipy drives, the device evaluates ----------------------------------
    def golden_sym_loss(self, range, data):
        neg_thr = 0 if self.one_sided_dist else -range
        return self.loss_fx(data, neg_thr, range)

    def golden_asym_shift_loss(self, shift, range, data):
        return self.loss_fx(data, -range + shift, range + shift)

    def golden_asym_range_loss(self, range, data):
        temp_delta = 2 * range / (2 ** self.quantizer.n_bits - 1)
        max_shift = temp_delta * self.max_int_skew
        result = minimize_scalar(self.golden_asym_shift_loss, args=(range, data),
                                 bounds=(-max_shift, max_shift), method='Bounded')
        return result.fun

    def _segments(self, data):
        if self._grouped(data):
            gs = data.shape[-1] // self.n_groups
            for g in range(self.n_groups):
                yield g, data[..., g * gs:(g + 1) * gs].contiguous()
            return
        for g in range(self.channel_groups):
            yield g, (data if (g == 0 and not self.per_channel) else data[g])

    def _expand_groups(self, data, xmin, xmax):
        if self._grouped(data):
            gs = data.shape[-1] // self.n_groups
            return xmin.repeat_interleave(gs), xmax.repeat_interleave(gs)
        return xmin, xmax

    def _golden_section_symmetric(self, data):
        xmin = torch.zeros(self.channel_groups)
        xmax = torch.zeros(self.channel_groups)
        for g, seg in self._segments(data):
            self.result = minimize_scalar(
                self.golden_sym_loss, args=seg,
                bounds=(0.01 * self.max_search_range, self.max_search_range), method='Bounded')
            xmax[g] = torch.tensor(self.result.x)
            xmin[g] = torch.tensor(0.0) if self.one_sided_dist else -xmax[g]
        xmin, xmax = self._expand_groups(data, xmin, xmax)
        self.current_xmax = xmax.to(data.device)
        self.current_xmin = xmin.to(data.device)

    def _golden_section_asymmetric(self, data):
        xmin = torch.zeros(self.channel_groups)
        xmax = torch.zeros(self.channel_groups)
        for g, seg in self._segments(data):
            self.result = minimize_scalar(
                self.golden_asym_range_loss, args=seg,
                bounds=(0.01 * self.max_search_range, self.max_search_range), method='Bounded')
            self.final_range = self.result.x
            temp_delta = 2 * self.final_range / (2 ** self.quantizer.n_bits - 1)
            max_shift = temp_delta * self.max_int_skew
            self.subresult = minimize_scalar(
                self.golden_asym_shift_loss, args=(self.final_range, seg),
                bounds=(-max_shift, max_shift), method='Bounded')
            self.final_shift = self.subresult.x
            xmax[g] = torch.tensor(self.final_range + self.final_shift)
            xmin[g] = torch.tensor(-self.final_range + self.final_shift)
        xmin, xmax = self._expand_groups(data, xmin, xmax)
        self.current_xmax = xmax.to(data.device)
        self.current_xmin = xmin.to(data.device)

    def _memoise(self, data):
        """Remember the thresholds just found for THIS tensor object in THIS state (identity through a weak reference +
        the version counter: an address can be recycled by another tensor, an object cannot)."""
        try:
            self._memo = (weakref.ref(data), data._version, (self.current_xmin, self.current_xmax))
        except RuntimeError:                 # inference tensor: no version counter, no memo
            self._memo = None

    def forward(self, data):
        memo = self._memo
        if memo is not None:
            # the golden-section search is a pure function of (input, search range): the thresholds a lock-step search
            # (golden_section_lockstep) found for THIS tensor are what running it again would return
            if memo[0]() is data and memo[1] == _hip_version(data) and self.opt_method == OptMethod.golden_section:
                self.current_xmin, self.current_xmax = memo[2]
                return self.current_xmin, self.current_xmax
            self._memo = None
        if self._loss_dev is None:
            if self.one_sided_dist is None:
                mn, _ = self._tensor_stats(data)
                self.one_sided_dist = bool(float(mn) >= 0)
            self._define_search_range(data)
        self.optimization_method(data)
        return self.current_xmin, self.current_xmax

    def reset(self):
        super().reset()
        self._loss_dev = None
        self._memo = None


class CrossEntropyEstimator(MSE_Estimator):
    """Same search, with -sum softmax(x) * log_softmax(Q(x)) as the loss (reference :493-502)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)

    def _launch_loss(self, be, data, rows, cand_dev, loss, per_row=False):
        # per_channel_loss only exists for signature compatibility upstream: one loss row
        be.xent_candidates(data, cand_dev, loss)

    def _rows(self, data):
        return 1


# --------------------------------------------------------------------------------------
# K independent golden-section searches in lock step
# --------------------------------------------------------------------------------------
def _bounded_brent(bounds, xatol=1e-5, maxiter=500):
    """scipy.optimize.minimize_scalar(method='Bounded') as a RESUMABLE search: a generator that yields the next abscissa
    and is sent the function value there; returns an OptimizeResult.  Restates scipy's `_minimize_scalar_bounded`
    (scipy/optimize/_optimize.py, 1.15: Brent's fminbound, the routine behind the reference's golden-section option,
    range_estimators.py:321, 429, 449, 458) expression by expression INCLUDING the numpy scalar types scipy's code
    produces on the way (np.abs / np.sign / np.maximum return numpy scalars, the function values are fp32 numpy
    scalars, the bounds Python floats: mixed expressions promote exactly as they do there), so the iterates are
    scipy's bit for bit -- pinned against scipy itself by tests/test_lockstep.py on random objectives."""
    from math import sqrt
    from scipy.optimize import OptimizeResult
    x1, x2 = bounds
    flag = 0
    sqrt_eps = sqrt(2.2e-16)
    golden_mean = 0.5 * (3.0 - sqrt(5.0))
    a, b = x1, x2
    fulc = a + golden_mean * (b - a)
    nfc, xf = fulc, fulc
    rat = e = 0.0
    x = xf
    fx = yield x
    num = 1
    fu = np.inf
    ffulc = fnfc = fx
    xm = 0.5 * (a + b)
    tol1 = sqrt_eps * np.abs(xf) + xatol / 3.0
    tol2 = 2.0 * tol1
    while np.abs(xf - xm) > (tol2 - 0.5 * (b - a)):
        golden = 1
        if np.abs(e) > tol1:                      # try a parabola through the three best points
            golden = 0
            r = (xf - nfc) * (fx - ffulc)
            q = (xf - fulc) * (fx - fnfc)
            p = (xf - fulc) * q - (xf - nfc) * r
            q = 2.0 * (q - r)
            if q > 0.0:
                p = -p
            q = np.abs(q)
            r = e
            e = rat
            if (np.abs(p) < np.abs(0.5 * q * r)) and (p > q * (a - xf)) and (p < q * (b - xf)):
                rat = (p + 0.0) / q
                x = xf + rat
                if ((x - a) < tol2) or ((b - x) < tol2):
                    si = np.sign(xm - xf) + ((xm - xf) == 0)
                    rat = tol1 * si
            else:
                golden = 1
        if golden:                                # golden-section step into the larger part
            e = a - xf if xf >= xm else b - xf
            rat = golden_mean * e
        si = np.sign(rat) + (rat == 0)
        x = xf + si * np.maximum(np.abs(rat), tol1)
        fu = yield x
        num += 1
        if fu <= fx:
            if x >= xf:
                a = xf
            else:
                b = xf
            fulc, ffulc = nfc, fnfc
            nfc, fnfc = xf, fx
            xf, fx = x, fu
        else:
            if x < xf:
                a = x
            else:
                b = x
            if (fu <= fnfc) or (nfc == xf):
                fulc, ffulc = nfc, fnfc
                nfc, fnfc = x, fu
            elif (fu <= ffulc) or (fulc == xf) or (fulc == nfc):
                fulc, ffulc = x, fu
        xm = 0.5 * (a + b)
        tol1 = sqrt_eps * np.abs(xf) + xatol / 3.0
        tol2 = 2.0 * tol1
        if num >= maxiter:
            flag = 1
            break
    if np.isnan(xf) or np.isnan(fx) or np.isnan(fu):
        flag = 2
    return OptimizeResult(fun=fx, status=flag, success=(flag == 0),
                          message={0: 'Solution found.', 1: 'Maximum number of function calls reached.',
                                   2: 'NaN result encountered.'}.get(flag, ''), x=xf, nfev=num, nit=num)


def _minimize_bounded(bounds, evaluate):
    """Drive `_bounded_brent` with an objective that is itself a generator (it may yield loss requests, or run a nested
    search): -> OptimizeResult."""
    g = _bounded_brent(bounds)
    x = next(g)
    while True:
        fx = yield from evaluate(x)
        try:
            x = g.send(fx)
        except StopIteration as stop:
            return stop.value


def _search_program(est):
    """The golden-section search of ONE estimator (one range per tensor) as a generator: yields (neg_thr, pos_thr)
    requests, is sent their fp32 losses, leaves the estimator in the state `_golden_section_symmetric /
    _golden_section_asymmetric` leave (reference range_estimators.py:296-327, 422-470) and returns (xmin, xmax)."""
    def loss(neg_thr, pos_thr):
        assert neg_thr or pos_thr, 'both thresholds falsy: the search range never contains it'
        value = yield (neg_thr, pos_thr)
        return value

    lo, hi = 0.01 * est.max_search_range, est.max_search_range
    if est._one_dimensional:
        def sym(r):
            return (yield from loss(0 if est.one_sided_dist else -r, r))
        est.result = yield from _minimize_bounded((lo, hi), sym)
        xmin, xmax = torch.zeros(est.channel_groups), torch.zeros(est.channel_groups)      # fp32 [1], as the sequential search
        xmax[0] = torch.tensor(est.result.x)
        xmin[0] = torch.tensor(0.0) if est.one_sided_dist else -xmax[0]
        return xmin, xmax

    def shift_search(r):
        temp_delta = 2 * r / (2 ** est.quantizer.n_bits - 1)
        max_shift = temp_delta * est.max_int_skew

        def shifted(shift):
            return (yield from loss(-r + shift, r + shift))
        return (yield from _minimize_bounded((-max_shift, max_shift), shifted))

    def range_loss(r):
        return (yield from shift_search(r)).fun
    est.result = yield from _minimize_bounded((lo, hi), range_loss)
    est.final_range = est.result.x
    est.subresult = yield from shift_search(est.final_range)
    est.final_shift = est.subresult.x
    xmin, xmax = torch.zeros(est.channel_groups), torch.zeros(est.channel_groups)
    xmax[0] = torch.tensor(est.final_range + est.final_shift)
    xmin[0] = torch.tensor(-est.final_range + est.final_shift)
    return xmin, xmax


def golden_section_lockstep(jobs):
    """Run the golden-section range searches of `jobs` = [(MSE_Estimator, tensor), ...] together, in lock step.

    Every search is a resumable program (`_search_program`: scipy's bounded Brent restated as a generator, nested for
    the asymmetric two-sided case).  Per ROUND the pending candidate of every live search is evaluated: one candidate-table
    upload, the launches of all tensors queued back to back, ONE device->host copy; each search is then sent the fp32 loss
    `loss_fx` would have computed for it alone -- same objective values, same iterates, same thresholds, bit for bit.
    Each estimator ends in the state `estimator(tensor)` leaves plus a memo of the tensor it saw, so that the estimating
    forward that follows finds its answer without launching anything.  Requirements (`lockstep_eligible`): plain
    MSE_Estimator, opt_method golden_section, one range per tensor, tensors on ONE device, calibration not sharded.
    Reference: range_estimators.py:296-327, 422-470 run once per weight tensor with a host round trip per loss evaluation
    (~20 per search; 102 searches under README.md:149-157).  -> {'searches', 'rounds', 'evaluations'}"""
    if not jobs:
        return {'searches': 0, 'rounds': 0, 'evaluations': 0}
    be = _hip.backend()
    originals = [data for _, data in jobs]           # the memo is keyed on the object the caller will pass again
    jobs = [(est, data.detach()) for est, data in jobs]
    device = jobs[0][1].device
    # ---- search ranges: every tensor's (min, max) with ONE host copy -------------------------------------------------------
    fresh = [k for k, (est, _) in enumerate(jobs) if est._loss_dev is None]
    if fresh:
        stats = torch.stack([torch.stack(be.minmax(jobs[k][1], 1, 1)) for k in fresh]).cpu()
        for row, k in zip(stats, fresh):
            est, data = jobs[k]
            if est.one_sided_dist is None:
                est.one_sided_dist = bool(float(row[0]) >= 0)
            est._define_search_range(data, stats=(row[0], row[1]))
    # prepared launches (shapes, pointers, one shared workspace resolved once) where the backend offers them
    prepared = be.mse_ordered_plans([data for _, data in jobs]) if hasattr(be, 'mse_ordered_plans') else None
    programs = {k: _search_program(est) for k, (est, _) in enumerate(jobs)}
    pending = {k: next(g) for k, g in programs.items()}            # every search starts with one evaluation
    rounds = evaluations = 0
    with torch.no_grad():
        while pending:
            slots = sorted(pending)
            cfgs = {(jobs[k][0].quantizer.n_bits, jobs[k][0].quantizer.symmetric, jobs[k][0].quantizer.eps,
                     jobs[k][0].quantizer.scale_domain) for k in slots}
            if len(cfgs) == 1:                   # the usual case (one recipe for all layers): ONE vectorised table build
                table = jobs[slots[0]][0]._cand_table([pending[k][0] for k in slots], [pending[k][1] for k in slots])
            else:
                table = np.concatenate([jobs[k][0]._cand_table([pending[k][0]], [pending[k][1]]) for k in slots], axis=0)
            cand = be.candidate_table(table, device)
            loss = be.zeros_f64((len(slots), 1), device)
            if prepared is not None:
                c0, l0, st = cand.data_ptr(), loss.data_ptr(), _hip._stream()
                for j, k in enumerate(slots):
                    be.mse_ordered_launch(prepared[0][k], c0 + 16 * j, l0 + 8 * j, st)
            else:
                for j, k in enumerate(slots):
                    be.mse_candidates_ordered(jobs[k][1], cand[j:j + 1], loss[j:j + 1], per_row=False)
            host = loss.cpu().numpy()                              # the round's one synchronisation
            rounds += 1
            evaluations += len(slots)
            nxt = {}
            for j, k in enumerate(slots):
                est, data = jobs[k]
                v = host[j, 0]
                v = v if data.dtype == torch.float64 else np.float32(v)      # exact: the cell holds one fp32 value
                try:
                    nxt[k] = programs[k].send(v)
                except StopIteration as stop:
                    xmin, xmax = stop.value
                    est.current_xmin, est.current_xmax = xmin.to(data.device), xmax.to(data.device)
                    est._memoise(originals[k])
            pending = nxt
    return {'searches': len(jobs), 'rounds': rounds, 'evaluations': evaluations}


def lockstep_eligible(est, data):
    return (type(est) is MSE_Estimator and est.opt_method == OptMethod.golden_section and not est.per_channel
            and not est._grouped(data) and torch.is_tensor(data) and data.dim() > 0 and data.numel() > 0
            and (data.is_cuda or getattr(_hip.backend(), 'accepts_cpu', False)) and not tq_dist.is_enabled())


RangeEstimatorMap = namedtuple('RangeEstimatorMap', ['value', 'cls'])


class RangeEstimators(Enum):
    current_minmax = RangeEstimatorMap(0, CurrentMinMaxEstimator)
    allminmax = RangeEstimatorMap(1, AllMinMaxEstimator)
    running_minmax = RangeEstimatorMap(2, RunningMinMaxEstimator)
    MSE = RangeEstimatorMap(3, MSE_Estimator)
    cross_entropy = RangeEstimatorMap(4, CrossEntropyEstimator)

    @property
    def cls(self):
        return self.value.cls

    @classmethod
    def list(cls):
        return [m.name for m in cls]
