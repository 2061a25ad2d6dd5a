"""Raw-RCCL communicator for the statistics exchange of sharded calibration (new).

`quantization.distributed` hands its collectives -- the MAX all-reduce of a `[-min | max]` statistics buffer, the SUM
all-reduce of fp64 candidate losses and of AdaRound's gradient -- to `RawRcclComm` instead of `torch.distributed` when a
raw communicator is active: one ctypes call into libtq_hip.so (`tq_comm_allreduce`, or the whole calibrating step as
`tq_calibrate_minmax_rccl`) which calls `ncclAllReduce` on the current HIP stream.  No c10d work objects, no watchdog
thread, no extra stream hop: ~3 us of host time per collective instead of ~37 us, and the launches capture into a
hipGraph like any other kernel.

Set-up needs ONE out-of-band exchange: the 128-byte ncclUniqueId made by rank 0.  It travels through a key-value
store -- the default process group's rendezvous store when torch.distributed is initialised (any backend: `gloo` is
enough, no c10d collective is issued), else a `TCPStore` on MASTER_ADDR / MASTER_PORT.  librccl itself is the one torch
already mapped (`torch/lib/librccl.so`), so the process never holds two RCCL runtimes.
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


class RawRcclComm:
    """One RCCL communicator over all ranks of the job, bound to this process's current device."""

    def __init__(self, rank=None, world=None, store=None, tag=None):
        global _generation
        if not torch.cuda.is_available():
            raise _hip.TQError('the raw RCCL exchange needs a GPU')
        self.lib = _hip.load_library()
        path = _librccl_path()
        _hip._check(self.lib.tq_comm_load(path.encode() if path else None), self.lib)
        self._store = None
        if store is None:
            store, srank, sworld = _default_store()
            self._store = store
            rank = srank if rank is None else rank
            world = sworld if world is None else world
        if rank is None or world is None:
            raise ValueError('rank / world are required with an explicit store')
        self.rank, self.world = int(rank), int(world)
        self.device = torch.device('cuda', torch.cuda.current_device())
        nb = int(self.lib.tq_comm_unique_id_bytes())
        # every rank creates its communicators in the same order, so a per-process counter names the exchange
        key = 'tq_rccl_uid/%s' % (tag if tag is not None else _generation)
        _generation += 1
        if self.rank == 0:
            uid = (C.c_ubyte * nb)()
            _hip._check(self.lib.tq_comm_get_unique_id(uid), self.lib)
            store.set(key, bytes(uid))
            raw = bytes(uid)
        else:
            raw = bytes(store.get(key))           # blocks until rank 0 has published it
        if len(raw) != nb:
            raise _hip.TQError(f'ncclUniqueId of {len(raw)} bytes, expected {nb}')
        buf = (C.c_ubyte * nb).from_buffer_copy(raw)
        comm = C.c_void_p()
        torch.cuda.synchronize()
        _hip._check(self.lib.tq_comm_init(buf, self.rank, self.world, C.byref(comm)), self.lib)
        self.handle = comm.value
        self.calls = 0
        self.version = int(self.lib.tq_comm_version())

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
            ok = ok and bool(torch.equal(self.allreduce_(v, MAX), want))
            d = (base.double() + r).contiguous()
            want = base.double() * w + sum(range(w))
            ok = ok and bool(torch.equal(self.allreduce_(d, SUM), want))
        flag = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
        self.allreduce_(flag, MIN)
        return bool(int(flag[0]))

    def close(self):
        if self.handle:
            torch.cuda.synchronize()
            self.lib.tq_comm_destroy(self.handle)
            self.handle = None
