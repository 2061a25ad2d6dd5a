"""P2P "mailbox" MAX all-reduce for the tiny statistics exchange of sharded calibration (new).

`quantization.distributed.sync_max_inplace` hands a `[-min | max]` buffer of <= 8 KB to `P2PMailbox.allreduce_max_`
instead of `torch.distributed.all_reduce` when a mailbox is active: ONE small kernel (tq_mailbox_allreduce_max: post
to the own mailbox, poll the peers' over xGMI) and no host bookkeeping, against ~37 us of host time per call through
c10d + RCCL (`profiles/r02/sharded_calibration_rccl1.json`).  Set-up: every rank allocates its mailbox inside
libtq_hip.so, the 64-byte IPC handles travel through `all_gather_object`, peers are mapped with hipIpcOpenMemHandle
(`HSA_ENABLE_IPC_MODE_LEGACY=0` must be set, as for RCCL on this driver).  `self_test()` compares the path with
`torch.distributed.all_reduce(MAX)` on known vectors; `quantization.distributed.enable` keeps RCCL if it fails.
The kernel's wait is bounded (default ~10 minutes); `timed_out()` is checked when calibration ends
(`quantization.distributed.check_exchange_health`), which raises instead of leaving NaN ranges behind.
"""
import ctypes as C

import torch
import torch.distributed as dist

from quantization import _hip


class P2PMailbox:
    def __init__(self, group=None, spin_budget=0):
        self.lib = _hip.load_library()
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.spin_budget = int(spin_budget)
        self.max_floats = int(self.lib.tq_mailbox_max_floats())
        self.device = torch.device('cuda', torch.cuda.current_device())
        hb = int(self.lib.tq_mailbox_handle_bytes())
        self.base, self._opened = None, []
        # Every set-up collective below runs on EVERY rank, whatever failed locally: a rank that skipped one would leave
        # its peers blocked in it (or paired with its next, unrelated collective).  Local failures are recorded and the
        # verdict is agreed at the end.
        err, handle = None, (C.c_ubyte * hb)()
        try:
            base = C.c_void_p()
            _hip._check(self.lib.tq_mailbox_alloc(C.byref(base), handle), self.lib)
            self.base = base.value
        except Exception as e:      # noqa: BLE001
            err = e
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(handle) if err is None else None, group=group)
        ptrs = []
        if err is None and all(h is not None for h in handles):
            try:
                for r, h in enumerate(handles):
                    if r == self.rank:
                        ptrs.append(self.base)
                        continue
                    buf = (C.c_ubyte * hb).from_buffer_copy(h)
                    peer = C.c_void_p()
                    _hip._check(self.lib.tq_mailbox_open(buf, C.byref(peer)), self.lib)
                    self._opened.append(peer.value)
                    ptrs.append(peer.value)
            except Exception as e:  # noqa: BLE001
                err = e
        elif err is None:
            err = RuntimeError('a peer could not allocate its mailbox')
        verdicts = [None] * self.world
        dist.all_gather_object(verdicts, err is None, group=group)     # doubles as the barrier: every mailbox is mapped
        if not all(verdicts):
            self._release()
            raise RuntimeError('P2P mailbox set-up failed on rank(s) %s%s' % (
                [r for r, v in enumerate(verdicts) if not v], '' if err is None else ': %s' % err))
        self.peers = torch.tensor(ptrs, dtype=torch.int64, device=self.device)     # void*[world] on the device
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.calls = 0

    def usable(self, buf):
        return (buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous() and 1 <= buf.numel() <= self.max_floats
                and buf.device == self.device)

    def allreduce_max_(self, buf):
        rc = self.lib.tq_mailbox_allreduce_max(buf.data_ptr(), buf.numel(), self.base, self.peers.data_ptr(), self.world,
                                               self.rank, self.status.data_ptr(), self.spin_budget, _hip._stream())
        _hip._check(rc, self.lib)
        self.calls += 1
        return buf

    def timed_out(self):
        """True if any call so far gave up waiting for a peer (host synchronisation)."""
        return bool(int(self.status[0]) & 1)

    def self_test(self, rounds=4):
        """The mailbox result equals torch.distributed's MAX all-reduce on rank-dependent vectors of several sizes."""
        g = torch.Generator(device=self.device).manual_seed(1234 + self.rank)
        ok = True
        budget, self.spin_budget = self.spin_budget, 500000        # fail fast (~1 s) here: the fallback is RCCL
        dist.barrier(group=self.group)
        for i in range(rounds):
            n = (2, 12, 1536, self.max_floats)[i % 4]
            v = torch.randn(n, device=self.device, generator=g) * (1 + self.rank)
            ref = v.clone()
            dist.all_reduce(ref, op=dist.ReduceOp.MAX, group=self.group)
            got = self.allreduce_max_(v.clone())
            ok = ok and bool(torch.equal(got, ref))
        self.spin_budget = budget
        ok = ok and not self.timed_out()
        flag = torch.tensor([1 if ok else 0], device=self.device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)          # all ranks agree on the verdict
        return bool(int(flag[0]))

    def _release(self):
        for p in self._opened:
            self.lib.tq_mailbox_close(p)
        self._opened = []
        if self.base:
            self.lib.tq_mailbox_free(self.base)
            self.base = None

    def close(self):
        torch.cuda.synchronize()
        try:
            dist.barrier(group=self.group)
        except Exception:       # noqa: BLE001  (process group may already be gone)
            pass
        self._release()
