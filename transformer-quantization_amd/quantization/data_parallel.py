"""Data-parallel QAT step: bucketed all-reduce of parameter and learnable-range gradients (new).

The reference trains on one device (reference utils/qat_utils.py:14-45 prepares the model, main.py:654 hands it to a
single-process Trainer); BASELINE configs[4] runs the same QAT forward/backward data-parallel over the 8 GPUs of a node.
One process per GPU holds a full replica; after `loss.backward()` the gradients of every trainable tensor -- weights,
biases and, after `learn_ranges()`, the quantizers' `_delta` / `_zero_float` / `x_min` / `x_max` parameters (reference
quantization/quantizers.py:284-288,346-349) -- are summed over the ranks and divided by the world size, which for a
mean-reduced loss over equal shards is exactly the gradient of the loss on the concatenated batch.

Design (xGMI is point-to-point: few, large collectives):

* gradients LIVE in flat buckets (`p.grad` is a view into one contiguous fp32 buffer per ~25 MB of parameters, filled in
  reverse registration order = roughly the order backward produces them), so there is no flatten / unflatten copy and the
  all-reduce runs in place;
* a bucket is reduced as soon as its last gradient has been accumulated (`register_post_accumulate_grad_hook`), on a
  second HIP stream that waits for the producing stream, so the exchange of layer L overlaps the backward of layer L-1;
  buckets are always launched in index order, whatever order the hooks fire in, so every rank issues the same sequence;
* transport: the raw RCCL communicator of `quantization/rccl.py` (`tq_comm_allreduce`; one ctypes call, hipGraph
  capturable -- the whole step including its collectives replays as one graph, `quantization.graphs.GraphedTrainStep`)
  when `quantization.distributed` has one, else `torch.distributed` (gloo in the CPU tests).
"""
import torch
import torch.distributed as dist

from quantization import distributed as tq_dist

DEFAULT_BUCKET_BYTES = 25 << 20


class GradientBuckets:
    """``gb = GradientBuckets(model.parameters()); gb.zero_(); loss.backward(); gb.finish(); optimizer.step()``

    ONE backward per step: a bucket is reduced when each of its gradients has been accumulated once since `zero_()`;
    micro-batch accumulation over several backward calls would have to suspend the exchange
    (`quantization.distributed.suspended()`) for all but the last of them.
    `average=True` divides the sum by the world size (mean-reduced loss, equal shards).  With no active exchange
    (`quantization.distributed` disabled or one rank without `force`) the object only provides the flat gradient storage
    and `finish()` is a no-op, so the same training loop runs on one GPU."""

    def __init__(self, params, bucket_bytes=DEFAULT_BUCKET_BYTES, average=True, overlap=True):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('GradientBuckets: no trainable parameter')
        self.average = bool(average)
        self.bucket_bytes = int(bucket_bytes)
        self._flats, self._bucket_of, self._pending0 = [], {}, []
        cur, cur_bytes, cur_key = [], 0, None
        plan = []
        for p in reversed(self.params):
            if p.dtype not in (torch.float32, torch.float64):
                raise TypeError(f'GradientBuckets: fp32 / fp64 master parameters only (got {p.dtype})')
            key = (p.device, p.dtype)
            nb = p.numel() * p.element_size()
            if cur and (key != cur_key or cur_bytes + nb > self.bucket_bytes):
                plan.append((cur_key, cur))
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
            cur_key = key
        if cur:
            plan.append((cur_key, cur))
        for k, ((device, dtype), ps) in enumerate(plan):
            flat = torch.zeros(sum(p.numel() for p in ps), device=device, dtype=dtype)
            off = 0
            for p in ps:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                self._bucket_of[id(p)] = k
            self._flats.append(flat)
            self._pending0.append(len(ps))
        self._pending = list(self._pending0)
        self._next = 0
        self._side = None
        dev = self._flats[0].device
        if overlap and dev.type == 'cuda':
            self._side = torch.cuda.Stream(device=dev)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self.launched = 0           # collectives issued so far (diagnostics / tests)

    # ---- bookkeeping ------------------------------------------------------------------------------------------------
    @property
    def n_buckets(self):
        return len(self._flats)

    def bucket_sizes(self):
        return [f.numel() * f.element_size() for f in self._flats]

    def zero_(self):
        """Start of a step (replaces optimizer.zero_grad(): the gradients must stay views of the buckets)."""
        for p in self.params:
            k = self._bucket_of[id(p)]
            if p.grad is None or p.grad.untyped_storage().data_ptr() != self._flats[k].untyped_storage().data_ptr():
                raise RuntimeError('a gradient was detached from its bucket (optimizer.zero_grad(set_to_none=True)?): '
                                   'use GradientBuckets.zero_() instead')
        for f in self._flats:
            f.zero_()
        self._pending = list(self._pending0)
        self._next = 0

    def _active(self):
        return tq_dist.is_enabled()

    def _on_grad(self, p):
        k = self._bucket_of[id(p)]
        self._pending[k] -= 1
        if self._active():
            while self._next < len(self._flats) and self._pending[self._next] <= 0:
                self._launch(self._next)
                self._next += 1

    def _reduce(self, flat):
        tq_dist.sync_sum(flat)          # raw RCCL on the current stream for device tensors, else torch.distributed
        if self.average:
            flat.mul_(1.0 / dist.get_world_size(tq_dist.group()))
        self.launched += 1

    def _launch(self, k):
        flat = self._flats[k]
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._side):
                self._reduce(flat)
        else:
            self._reduce(flat)

    def finish(self):
        """After backward: reduce the buckets whose hooks did not all fire (parameters unused in this step), then make
        the current stream wait for the exchange.  Every rank reaches the same sequence of collectives."""
        if not self._active():
            return
        while self._next < len(self._flats):
            self._launch(self._next)
            self._next += 1
        if self._side is not None:
            torch.cuda.current_stream(self._flats[0].device).wait_stream(self._side)

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_parameters(module, root=0):
    """Every replica starts from rank `root`'s parameters and buffers (state_dict order)."""
    if not tq_dist.is_enabled():
        return
    raw = tq_dist.raw_comm()
    for t in module.state_dict().values():
        if not torch.is_tensor(t) or t.numel() == 0:
            continue
        if raw is not None and raw.usable(t):
            raw.broadcast_(t, root)
        else:
            dist.broadcast(t, src=root, group=tq_dist.group())


def train_step(module, loss_fn, optimizer, buckets, inputs, targets=()):
    """One eager data-parallel QAT iteration on this rank's shard -> detached local loss."""
    buckets.zero_()
    loss = loss_fn(module(*inputs), *targets)
    loss.backward()
    buckets.finish()
    optimizer.step()
    return loss.detach()
