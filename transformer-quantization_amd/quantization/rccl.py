"""Raw-RCCL communicator for the statistics exchange of sharded calibration (new).

`quantization.distributed` hands its collectives -- the MAX all-reduce of a `[-min | max]` statistics buffer, the SUM
all-reduce of fp64 candidate losses and of AdaRound's gradient -- to `RawRcclComm` instead of `torch.distributed` when a
raw communicator is active: one ctypes call into libtq_hip.so (`tq_comm_allreduce`, or the whole calibrating step as
`tq_calibrate_minmax_rccl`) which calls `ncclAllReduce` on the current HIP stream.  No c10d work objects, no watchdog
thread, no extra stream hop: ~3 us of host time per collective instead of ~37 us, and the launches capture into a
hipGraph like any other kernel.

Set-up goes through a key-value store -- the default process group's rendezvous store when torch.distributed is
initialised (any backend: `gloo` is enough, no c10d collective is issued), else a `TCPStore` on MASTER_ADDR /
MASTER_PORT: the 128-byte ncclUniqueId made by rank 0 travels through it, and so do the two agreement rounds that make
bring-up a two-phase commit (`RawRcclComm.__init__`: a failure on one rank is a failure on all).  librccl itself is the
one torch already mapped (`torch/lib/librccl.so`), so the process never holds two RCCL runtimes.
"""
import ctypes as C
import os

import torch

from quantization import _hip

F32, F64, I32, U8 = 0, 1, 2, 3
MAX, SUM, MIN = 0, 1, 2
_DTYPES = {torch.float32: F32, torch.float64: F64, torch.int32: I32, torch.uint8: U8}
_generation = 0


def _librccl_path():
    env = os.environ.get('TQ_RCCL_LIB')
    if env:
        return env
    cand = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
    return cand if os.path.exists(cand) else None


def _default_store():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        from torch.distributed import distributed_c10d as c10d
        return c10d._get_default_store(), dist.get_rank(), dist.get_world_size()
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    host, port = os.environ.get('MASTER_ADDR', '127.0.0.1'), int(os.environ.get('MASTER_PORT', '29500'))
    # (a second store next to torchrun's agent store: one port above the rendezvous port)
    port = int(os.environ.get('TQ_RCCL_STORE_PORT', port + 1))
    from datetime import timedelta
    store = dist.TCPStore(host, port, world, is_master=(rank == 0), timeout=timedelta(seconds=300),
                          wait_for_workers=False)
    return store, rank, world


class RawSetupFailed(_hip.TQError):
    """Bring-up of the raw communicator was rejected by the agreement round: raised on EVERY rank with the same list of
    failing ranks (never on a subset), so the callers' fallback -- torch.distributed stays in charge -- is taken by all
    ranks together."""

    def __init__(self, phase, failed):
        self.phase, self.failed = phase, list(failed)
        super().__init__('raw RCCL set-up failed in phase %r on rank(s) %s' % (
            phase, ', '.join('%d (%s)' % (r, m or 'no message') for r, m in failed)))


def agree(store, key, rank, world, ok, message='', timeout_s=None):
    """One agreement round through the key-value store: every rank publishes its local verdict under `key/<rank>` and
    reads the verdicts of all ranks.  -> [(rank, message) of every rank that reported a failure] (empty: go ahead).
    No collective is involved, so the round completes whatever state the device-side transports are in; a rank that never
    publishes (dead process) makes the others time out of `store.wait` with an exception instead of hanging for ever."""
    store.set('%s/%d' % (key, rank), (b'1' if ok else b'0') + str(message).encode('utf-8', 'replace')[:400])
    keys = ['%s/%d' % (key, r) for r in range(world)]
    if timeout_s is not None:
        from datetime import timedelta
        store.wait(keys, timedelta(seconds=float(timeout_s)))
    failed = []
    for r, k in enumerate(keys):
        v = bytes(store.get(k))
        if v[:1] != b'1':
            failed.append((r, v[1:].decode('utf-8', 'replace')))
    return failed


class RawRcclComm:
    """One RCCL communicator over all ranks of the job, bound to this process's current device.

    Bring-up is a two-phase commit over the rendezvous store (`agree`), so that a LOCAL failure on one rank (library
    missing, bad id, device error) never leaves the ranks on different transports or some of them blocked inside
    ncclCommInitRank.  (A rank that dies inside ncclCommInitRank itself still leaves its peers to RCCL's own timeout.)

    1. *prepare* (local, no peer involved): bind librccl (`tq_comm_load`), rank 0 makes the unique id.  Every rank then
       publishes ok / failed and reads everybody's verdict; unless all are ok, all ranks raise `RawSetupFailed` -- nobody
       has entered a collective call yet.  1b: every rank fetches the id, checks its length and drains its device, and a
       second round agrees on THAT: phase 2 then contains nothing that can fail locally before the collective.
    2. *commit*: `ncclCommInitRank` on every rank, the communicator's own rank / size are compared with the job's, and
       a second round agrees the outcome; on any failure every rank that holds a communicator aborts it
       (`tq_comm_abort`, which does not wait for peers) and all raise `RawSetupFailed`.

    `lib` / `device`: test doubles for the CPU tests of the protocol (tests/test_rccl_raw.py); the product passes neither.
    """

    def __init__(self, rank=None, world=None, store=None, tag=None, lib=None, device=None, agree_timeout_s=300.0):
        global _generation
        # every rank creates its communicators in the same order, so a per-process counter names the exchange; it
        # advances BEFORE anything can fail, so a rejected set-up does not desynchronise the ranks' next attempt
        gen = _generation
        _generation += 1
        self.handle, self.lib, self._store = None, None, None
        if store is None:
            store, srank, sworld = _default_store()
            self._store = store
            rank = srank if rank is None else rank
            world = sworld if world is None else world
        if rank is None or world is None:
            raise ValueError('rank / world are required with an explicit store')
        self.rank, self.world = int(rank), int(world)
        key = 'tq_rccl/%s' % (tag if tag is not None else gen)

        # ---- phase 1: prepare (local) -------------------------------------------------------------------------------
        err, uid, nb = None, b'', 0
        try:
            if lib is None and not torch.cuda.is_available():
                raise _hip.TQError('the raw RCCL exchange needs a GPU')
            self.lib = lib if lib is not None else _hip.load_library()
            self.device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
            path = _librccl_path()
            _hip._check(self.lib.tq_comm_load(path.encode() if path else None), self.lib)
            nb = int(self.lib.tq_comm_unique_id_bytes())
            if self.rank == 0:
                buf = (C.c_ubyte * nb)()
                _hip._check(self.lib.tq_comm_get_unique_id(buf), self.lib)
                uid = bytes(buf)
        except Exception as e:      # noqa: BLE001 -- reported to every rank below
            err = e
        if self.rank == 0:
            store.set(key + '/uid', uid)            # (empty when rank 0 failed: nobody reads it then)
        failed = agree(store, key + '/prepare', self.rank, self.world, err is None, repr(err) if err else '', agree_timeout_s)
        if failed:
            raise RawSetupFailed('prepare', failed) from err

        # ---- phase 1b: everything else that can fail LOCALLY (id fetch and length, draining the device) ------------------
        # agreed on separately so that phase 2 holds nothing but the collective: a rank that fails here has not entered
        # ncclCommInitRank, and no peer enters it before every rank has reported
        buf = None
        try:
            raw = uid if self.rank == 0 else bytes(store.get(key + '/uid'))
            if len(raw) != nb:
                raise _hip.TQError(f'ncclUniqueId of {len(raw)} bytes, expected {nb}')
            buf = (C.c_ubyte * nb).from_buffer_copy(raw)
            if self.device.type == 'cuda':
                torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001
            err = e
        failed = agree(store, key + '/fetched', self.rank, self.world, err is None, repr(err) if err else '', agree_timeout_s)
        if failed:
            raise RawSetupFailed('prepare', failed) from err

        # ---- phase 2: commit (collective) ---------------------------------------------------------------------------
        # (a rank that dies INSIDE ncclCommInitRank still leaves its peers to RCCL's own timeout: nothing a caller can do)
        try:
            comm = C.c_void_p()
            _hip._check(self.lib.tq_comm_init(buf, self.rank, self.world, C.byref(comm)), self.lib)
            self.handle = comm.value
            got = self.rank_world()
            if got != (self.rank, self.world):
                raise _hip.TQError('communicator reports rank %d of %d, the job says %d of %d' % (got + (self.rank, self.world)))
        except Exception as e:      # noqa: BLE001
            err = e
        failed = agree(store, key + '/commit', self.rank, self.world, err is None, repr(err) if err else '', agree_timeout_s)
        if failed:
            self.abort()
            raise RawSetupFailed('commit', failed) from err
        self.calls = 0
        self.version = int(self.lib.tq_comm_version())

    def rank_world(self):
        """(rank, size) as the COMMUNICATOR reports them (ncclCommUserRank / ncclCommCount), not the environment."""
        r, w = C.c_int(-1), C.c_int(-1)
        _hip._check(self.lib.tq_comm_rank_world(self.handle, C.byref(r), C.byref(w)), self.lib)
        return int(r.value), int(w.value)

    def usable(self, t):
        return (t.is_cuda and t.device == self.device and t.is_contiguous() and t.dtype in _DTYPES and self.handle)

    def allreduce_(self, t, op):
        """In place on the current stream; returns `t`."""
        rc = self.lib.tq_comm_allreduce(self.handle, t.data_ptr(), t.numel(), _DTYPES[t.dtype], op, _hip._stream())
        _hip._check(rc, self.lib)
        self.calls += 1
        return t

    def broadcast_(self, t, root=0):
        rc = self.lib.tq_comm_broadcast(self.handle, t.data_ptr(), t.numel(), _DTYPES[t.dtype], int(root), _hip._stream())
        _hip._check(rc, self.lib)
        self.calls += 1
        return t

    def self_test(self):
        """Known rank-dependent vectors through MAX / SUM on fp32 and fp64; every rank checks against the closed form
        and the verdicts are MIN-reduced over the communicator itself, so all ranks agree."""
        dev, r, w = self.device, self.rank, self.world
        ok = True
        for n in (2, 12, 1536, 101):
            base = torch.arange(n, device=dev, dtype=torch.float32)
            v = (base * (1 + r) - 3.0 * r).contiguous()
            want = torch.stack([base * (1 + q) - 3.0 * q for q in range(w)]).max(0).values
            got = self.allreduce_(v, MAX)            # the collective is issued whatever the verdict so far: a rank
            ok = bool(torch.equal(got, want)) and ok     # that skipped one would pair its next call with a peer's this one
            d = (base.double() + r).contiguous()
            want = base.double() * w + sum(range(w))
            got = self.allreduce_(d, SUM)
            ok = bool(torch.equal(got, want)) and ok
        flag = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
        self.allreduce_(flag, MIN)
        return bool(int(flag[0]))

    def latency_us(self, n_bytes, reps=200, warmup=20):
        """Median host-observed time of ONE in-place fp32 SUM all-reduce of `n_bytes`, issued back to back on the current
        stream and bracketed by HIP events (device time per collective, launch gaps included) -- the per-exchange latency
        the sharded calibrating forward pays 161 times (bench.py `calibration_model.exchange_latency_us`)."""
        n = max(1, int(n_bytes) // 4)
        buf = torch.zeros(n, device=self.device, dtype=torch.float32)
        for _ in range(warmup):
            self.allreduce_(buf, SUM)
        torch.cuda.synchronize()
        samples = []
        chunk = 10
        for _ in range(max(1, reps // chunk)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(chunk):
                self.allreduce_(buf, SUM)
            b.record()
            b.synchronize()
            samples.append(a.elapsed_time(b) * 1e3 / chunk)
        samples.sort()
        return samples[len(samples) // 2]

    def abort(self):
        """Tear the communicator down without waiting for the peers (rejected set-up, peer gone)."""
        if self.handle:
            try:
                self.lib.tq_comm_abort(self.handle)
            finally:
                self.handle = None

    def close(self):
        if self.handle:
            if self.device.type == 'cuda':
                torch.cuda.synchronize()
            self.lib.tq_comm_destroy(self.handle)
            self.handle = None
