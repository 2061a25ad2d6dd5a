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

Messages are tiny, so the collectives are latency-bound: one fused buffer per call, no barrier, issued on
the current stream.  Transport, in order of preference:

1. the RAW RCCL communicator of ``quantization/rccl.py`` (``tq_comm_allreduce`` / the one-call sharded step
   ``tq_calibrate_minmax_rccl`` inside libtq_hip.so): no c10d in the data path, hipGraph-capturable; the default
   for device tensors when the process group's backend is ``nccl`` (RCCL on ROCm);
2. ``torch.distributed`` (``nccl`` == RCCL, or ``gloo`` in the CPU tests and for host tensors);
3. opt-in: the P2P mailbox kernel for buffers <= 8 KB (``quantization/mailbox.py``).
"""
import torch
import torch.distributed as dist

import logging
import os

_group = None
_enabled = False
_force = False     # run the collectives even in a 1-rank group (RCCL smoke tests on a 1-GPU box)
_mailbox = None    # quantization.mailbox.P2PMailbox when the P2P exchange is active
_raw = None        # quantization.rccl.RawRcclComm when the raw-RCCL exchange is active
_stats = {'minmax_calls': 0, 'sum_calls': 0, 'bytes': 0, 'mailbox_calls': 0, 'raw_rccl_calls': 0}
logger = logging.getLogger('tq.distributed')


def _want_raw(group, raw):
    if raw is None:
        env = os.environ.get('TQ_DIST_RAW_RCCL')
        if env is not None:
            raw = env == '1'
        else:
            raw = dist.get_backend(group) == 'nccl'      # the ranks own one device each: RCCL is usable
    if not raw or not torch.cuda.is_available():
        return False
    # the raw communicator spans the whole job: only for the default (world) group
    return group is None or dist.get_world_size(group) == dist.get_world_size()


def enable(group=None, force=False, mailbox=None, raw=None):
    """Turn on statistic all-reduce (requires an initialised default process group).  A 1-rank group
    skips the collectives unless `force` is set.

    raw: True / False / None (= environment TQ_DIST_RAW_RCCL, default: on when the group's backend is `nccl`): device
    tensors are exchanged on a raw RCCL communicator owned by libtq_hip.so (quantization/rccl.py) -- no c10d in the data
    path.  Bring-up is agreed by all ranks through the rendezvous store before and after ncclCommInitRank (a failure on
    one rank makes EVERY rank fall back), and the communicator is self-tested; if the test fails, torch.distributed
    stays in charge on every rank (the verdict is reduced over all ranks).
    mailbox: True / False / None (= environment TQ_DIST_MAILBOX, default off): exchange the <= 8 KB min/max buffers of
    the fused calibration step through the P2P mailbox kernel (quantization/mailbox.py) instead of an all-reduce.
    The path is self-tested against the all-reduce here; if the set-up or the test fails on any rank, it stays off."""
    global _group, _enabled, _force, _mailbox, _raw
    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError('torch.distributed is not initialised')
    _group, _enabled, _force = group, True, bool(force)
    if mailbox is None:
        mailbox = os.environ.get('TQ_DIST_MAILBOX', '0') == '1'
    _close_transports()
    active = force or dist.get_world_size(group) > 1
    if active and _want_raw(group, raw):
        from quantization import rccl
        try:
            # two-phase commit over the rendezvous store: either every rank holds a communicator afterwards or every
            # rank gets RawSetupFailed with the same verdict -- a local failure never splits the transports
            comm = rccl.RawRcclComm()
        except rccl.RawSetupFailed as e:
            if raw:                 # explicitly requested: do not hide it (raised on every rank alike)
                raise
            logger.warning('raw RCCL exchange unavailable (%s): statistics go through torch.distributed', e)
            comm = None
        if comm is not None:
            if comm.self_test():
                _raw = comm
            else:
                logger.warning('raw RCCL self-test failed: statistics go through torch.distributed')
                comm.close()
    if mailbox and torch.cuda.is_available() and active:
        box = None
        try:
            from quantization.mailbox import P2PMailbox
            box = P2PMailbox(group)             # (its set-up collectives run on every rank, whatever fails locally)
            ok = box.self_test()
        except Exception as e:      # noqa: BLE001 -- a verdict all ranks reached together: stay on the all-reduce
            logger.warning('P2P mailbox unavailable (%s): statistics go through the all-reduce', e)
            ok = False
        if ok:
            _mailbox = box
        elif box is not None:
            logger.warning('P2P mailbox self-test failed: statistics go through the all-reduce')
            try:
                box.close()
            except Exception:       # noqa: BLE001
                pass


def _close_transports():
    global _mailbox, _raw
    if _mailbox is not None:
        try:
            _mailbox.close()
        finally:
            _mailbox = None
    if _raw is not None:
        try:
            _raw.close()
        finally:
            _raw = None


def check_exchange_health():
    """Raise if the P2P mailbox gave up waiting for a peer at any point (the affected statistics are NaN).  Called when
    calibration ends (utils.pass_data_for_range_estimation, disable()); costs one host synchronisation."""
    if _mailbox is not None and _mailbox.timed_out():
        raise RuntimeError('P2P mailbox exchange timed out waiting for a peer during sharded calibration: the ranges '
                           'estimated since are invalid (NaN).  Re-run with TQ_DIST_MAILBOX=0 (RCCL waits for ever).')


def disable():
    global _group, _enabled, _force
    try:
        check_exchange_health()
    finally:
        _close_transports()
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


def raw_comm():
    """The active raw RCCL communicator (quantization.rccl.RawRcclComm) or None."""
    return _raw


def raw_comm_for(x):
    """The raw communicator if the fused one-call sharded step can run on `x` (a device tensor of this rank's GPU)."""
    if _raw is not None and x.is_cuda and x.device == _raw.device:
        return _raw
    return None


def count_raw_exchange(n_bytes):
    _stats['raw_rccl_calls'] += 1
    _stats['minmax_calls'] += 1
    _stats['bytes'] += n_bytes


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
    elif _raw is not None and _raw.usable(buf):
        _raw.allreduce_(buf, 0)                 # rccl.MAX
        _stats['raw_rccl_calls'] += 1
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
    if _raw is not None and _raw.usable(t):
        _raw.allreduce_(t, 1)                   # rccl.SUM
        _stats['raw_rccl_calls'] += 1
    else:
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
