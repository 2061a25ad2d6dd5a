"""Sharded calibration: all-reduce of per-rank range statistics over RCCL / xGMI.

The reference has no distributed code at all; this module is new.  Calibration
batches are split along the batch dimension across ranks (one process per GPU, identical weights).
Every range estimator computes its statistic on the local shard and calls one of the two hooks
below *inline* -- the freshly estimated range is needed to quantize the tensor that feeds the next
layer, so the exchange cannot be deferred to the end of the forward:

* ``sync_minmax``: min and max are fused into ONE ``MAX`` all-reduce on ``[-min ; max]``
  (2 floats per tensor, 2*768 per-embedding).  min/max are associative, so the result is
  bit-identical to the single-rank statistic of the concatenated batch; the EMA / all-time update
  is then applied identically on every rank.
* ``sync_sum``: ``SUM`` all-reduce of the fp64 candidate-loss vector of the MSE / cross-entropy
  search (808 B for the 1-D grid, 103 KB for the 2-D grid); the argmin is computed redundantly.

Messages are tiny, so the collectives are latency-bound: one fused buffer per call, no barrier,
issued on the current stream through ``torch.distributed`` (backend ``nccl`` == RCCL on ROCm;
``gloo`` in the CPU tests).
"""
import torch
import torch.distributed as dist

import logging
import os

_group = None
_enabled = False
_force = False     # run the collectives even in a 1-rank group (RCCL smoke tests on a 1-GPU box)
_mailbox = None    # quantization.mailbox.P2PMailbox when the P2P exchange is active
_stats = {'minmax_calls': 0, 'sum_calls': 0, 'bytes': 0, 'mailbox_calls': 0}
logger = logging.getLogger('tq.distributed')


def enable(group=None, force=False, mailbox=None):
    """Turn on statistic all-reduce (requires an initialised default process group).  A 1-rank group
    skips the collectives unless `force` is set.

    mailbox: True / False / None (= environment TQ_DIST_MAILBOX, default off): exchange the <= 8 KB min/max buffers of
    the fused calibration step through the P2P mailbox kernel (quantization/mailbox.py) instead of an RCCL all-reduce.
    The path is self-tested against RCCL here; if the set-up or the test fails on any rank, RCCL stays in charge."""
    global _group, _enabled, _force, _mailbox
    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError('torch.distributed is not initialised')
    _group, _enabled, _force = group, True, bool(force)
    if mailbox is None:
        mailbox = os.environ.get('TQ_DIST_MAILBOX', '0') == '1'
    if _mailbox is not None:
        _mailbox.close()
        _mailbox = None
    if mailbox and torch.cuda.is_available() and (force or dist.get_world_size(group) > 1):
        ok, box = True, None
        try:
            from quantization.mailbox import P2PMailbox
            box = P2PMailbox(group)
            ok = box.self_test()
        except Exception as e:      # noqa: BLE001 -- any set-up problem (IPC refused, library missing): stay on RCCL
            logger.warning('P2P mailbox unavailable (%s): statistics go through torch.distributed', e)
            ok = False
        if ok:
            _mailbox = box
        elif box is not None:
            logger.warning('P2P mailbox self-test failed: statistics go through torch.distributed')
            try:
                box.close()
            except Exception:       # noqa: BLE001
                pass


def disable():
    global _group, _enabled, _force, _mailbox
    if _mailbox is not None:
        try:
            _mailbox.close()
        finally:
            _mailbox = None
    _group, _enabled, _force = None, False, False


class suspended:
    """``with tq_dist.suspended(): ...`` -- the statistics / gradient exchanges are off inside the block (the process
    group stays): for work that is partitioned by OBJECT rather than by sample, e.g. layer-parallel AdaRound, where
    every rank runs complete, independent per-layer problems and only the results travel."""

    def __enter__(self):
        global _enabled
        self._was = _enabled
        _enabled = False
        return self

    def __exit__(self, *exc):
        global _enabled
        _enabled = self._was
        return False


def group():
    return _group


def mailbox_active():
    return _mailbox is not None


def mailbox_for(n_floats):
    """The active P2P mailbox if a buffer of `n_floats` fp32 values fits it, else None (RCCL path)."""
    if _mailbox is not None and n_floats <= _mailbox.max_floats:
        return _mailbox
    return None


def count_mailbox_exchange(n_bytes):
    _stats['mailbox_calls'] += 1
    _stats['minmax_calls'] += 1
    _stats['bytes'] += n_bytes


def is_enabled():
    return _enabled and dist.is_initialized() and (_force or dist.get_world_size(_group) > 1)


def stats():
    return dict(_stats)


def sync_minmax(mn, mx):
    """-> (min, max) over all ranks; one MAX all-reduce of cat(-min, max)."""
    if not is_enabled():
        return mn, mx
    shape = mn.shape
    buf = torch.cat([(-mn).reshape(-1), mx.reshape(-1)])
    # the same exchange as the fused step's (mailbox or RCCL): a rank on the layered path -- e.g. one whose shard is
    # empty -- must meet its peers in the same collective
    sync_max_inplace(buf)
    n = buf.numel() // 2
    return (-buf[:n]).reshape(shape), buf[n:].reshape(shape)


def sync_max_inplace(buf):
    """MAX all-reduce of a `[-min | max]` statistics buffer the kernel wrote (tq_calibrate_stats), in place: the
    whole exchange of a calibrating call is this one collective -- no cat / neg / slice launches around it."""
    if _mailbox is not None and _mailbox.usable(buf):
        _mailbox.allreduce_max_(buf)
        _stats['mailbox_calls'] += 1
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=_group)
    _stats['minmax_calls'] += 1
    _stats['bytes'] += buf.numel() * buf.element_size()
    return buf


def sync_sum(t):
    """-> element-wise sum over all ranks (fp64 candidate losses, AdaRound gradients)."""
    if not is_enabled():
        return t
    t = t.contiguous()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_group)
    _stats['sum_calls'] += 1
    _stats['bytes'] += t.numel() * t.element_size()
    return t


def shard_batch(x, dim=0):
    """The slice of a calibration batch this rank owns."""
    if not is_enabled():
        return x
    ws, rk = dist.get_world_size(_group), dist.get_rank(_group)
    n = x.shape[dim]
    per = (n + ws - 1) // ws
    return x.narrow(dim, min(rk * per, n), max(0, min(per, n - rk * per)))
