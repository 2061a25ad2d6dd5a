"""hipGraph capture of quantized forwards (new; the reference has no counterpart).

A fixed-range forward makes no host-side decision and no host synchronisation, and with
``options.INPLACE_CALIBRATION_STATE`` the same holds for a calibrating forward, so either can be recorded
once with ``torch.cuda.graph`` and replayed per batch: no Python, no allocator, back-to-back launches
(BERT-base W8A8, B=8, T=128 on MI355X: 6.0 -> 3.3 ms fixed-range, 6.9 -> 4.8 ms calibrating, 3.0 -> 1.0 ms
with the integer fast paths).
"""
import torch

from quantization import _hip, options


class CaptureRefused(RuntimeError):
    """Raised instead of recording a graph that could not be replayed safely."""


def _refuse_c10d_exchange(what):
    """A capture whose collectives would go through torch.distributed's `nccl` backend is REFUSED.

    c10d wraps every collective in work objects that its watchdog thread polls with event queries; such a query while a
    capture is open invalidates the capture or aborts the process (round 2's red test), and the work objects recorded
    during capture are not re-created on replay.  The raw communicator (`quantization/rccl.py`: `ncclAllReduce` called
    from libtq_hip.so on the capturing stream, no c10d object involved) is the supported transport for captured sharded
    calibration / data-parallel steps: `quantization.distributed.enable(raw=True)`."""
    from quantization import distributed as tq_dist
    if tq_dist.is_enabled() and tq_dist.raw_comm() is None:
        raise CaptureRefused(
            f'{what}: the statistics / gradient exchange is active and goes through torch.distributed; collectives '
            'issued by c10d cannot be captured into a hipGraph.  Enable the raw RCCL transport first '
            '(quantization.distributed.enable(raw=True), backend `nccl`), or capture with the exchange off '
            '(`with quantization.distributed.suspended(): ...`).')


def _quiesce():
    """Nothing of this process may be in flight when a capture opens.  Wait for the device; when a c10d `nccl` process
    group exists (it may: barriers and the like outside the captured region), also give its watchdog thread one polling
    period to retire the (now complete) work objects of eager collectives: an event query from that thread while a
    GLOBAL-mode capture is open aborts the process, which is why the captures below also use
    `capture_error_mode='thread_local'` (only the capturing thread's calls are policed).  Collectives INSIDE the
    captured region must come from the raw-RCCL exchange (quantization/rccl.py), see `_refuse_c10d_exchange`."""
    torch.cuda.synchronize()
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl':
            import time
            time.sleep(0.25)
            torch.cuda.synchronize()
    except Exception:       # noqa: BLE001 -- best effort: the thread-local capture mode is the actual guard
        pass


_CACHE_ATTRS = ('cached_params', '_int8_cache', '_int8_stair', '_qparam_cache', '_stacked_i8_cache')


def _tensors_in(obj, out, depth=0):
    if torch.is_tensor(obj):
        out.append(obj)
    elif isinstance(obj, dict) and depth < 4:
        for v in obj.values():
            _tensors_in(v, out, depth + 1)
    elif isinstance(obj, (tuple, list)) and depth < 4:
        for v in obj:
            _tensors_in(v, out, depth + 1)


def derived_cache_tensors(module):
    """Every tensor the modules below `module` hold in a derived cache right now (the reference's cached quantized
    parameters, int8 weight indices and row sums, stacked Q|K|V operands, NoNorm parameters, GELU staircase tables).  A
    recorded graph reads them by ADDRESS: the graph object keeps these references so that a cache that is rebuilt later
    (another range state, `options.invalidate_derived_caches()`, `.train()`) cannot hand the memory a replay still reads
    back to the allocator."""
    out = []
    for m in module.modules():
        for attr in _CACHE_ATTRS:
            _tensors_in(m.__dict__.get(attr), out)
    return out


class GraphedForward:
    """``g = GraphedForward(model, example_ids); logits = g(ids)``.

    * inputs are copied into static buffers of the example's shape / dtype (shapes must not change);
    * the returned tensors are the graph's static outputs: clone them to keep a result across calls;
    * `restore_state=True` snapshots every buffer of the module before the warm-up iterations and puts the
      values back after capture, so the warm-up does not count as extra batches for running min/max (EMA)
      estimators when a CALIBRATING forward is captured (requires options.INPLACE_CALIBRATION_STATE, and
      one eager batch before capture so that every state buffer exists).
    """

    def __init__(self, module, *example_inputs, warmup=2, restore_state=True, no_grad=True):
        if not all(torch.is_tensor(t) and t.is_cuda for t in example_inputs):
            raise ValueError('GraphedForward needs ROCm tensors as example inputs')
        _refuse_c10d_exchange('GraphedForward')
        self.module = module
        self.static_inputs = tuple(t.clone() for t in example_inputs)
        self._no_grad = no_grad
        snap = {k: v.clone() for k, v in module.state_dict().items()} if restore_state else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):          # allocator pools, workspaces, ticket words, caches
                self._run()
        torch.cuda.current_stream().wait_stream(side)
        _quiesce()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
            self.static_outputs = self._run()
        self._cache_refs = derived_cache_tensors(module)       # read by address in every replay: see derived_cache_tensors
        if snap is not None:
            live = module.state_dict()
            if live.keys() != snap.keys():
                raise RuntimeError('module state changed shape during capture: run one eager batch first')
            for k, v in live.items():
                if v.shape != snap[k].shape:
                    raise RuntimeError(f'buffer {k} was re-allocated during capture (enable '
                                       'options.INPLACE_CALIBRATION_STATE for calibrating forwards)')
                v.copy_(snap[k])

    def _run(self):
        if self._no_grad:
            with torch.no_grad():
                return self.module(*self.static_inputs)
        return self.module(*self.static_inputs)

    def __call__(self, *inputs):
        if len(inputs) != len(self.static_inputs):
            raise ValueError(f'expected {len(self.static_inputs)} inputs')
        for dst, src in zip(self.static_inputs, inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError(f'input shape / dtype changed: captured {tuple(dst.shape)} {dst.dtype}, '
                                 f'got {tuple(src.shape)} {src.dtype}')
            dst.copy_(src, non_blocking=True)
        _hip.raise_deferred()          # what an EARLIER replay could only flag (token ids outside the vocabulary): no sync
        self.graph.replay()
        return self.static_outputs


class GraphedTrainStep:
    """One QAT iteration -- zero_grad, forward, loss, backward, optimizer.step() -- recorded once and replayed per batch.

    ``step = GraphedTrainStep(model, loss_fn, optimizer, (ids,), (labels,)); loss = step((ids,), (labels,))``

    A QAT step with fixed or learnable ranges makes no host-side decision either: the fake-quant forward, its
    straight-through backward (`tq_fake_quant_bwd`, deterministic block-partial reductions for the range gradients),
    the integer Linear under autograd and a capturable optimizer are all plain launches on the current stream.  Eager
    mode pays ~25 us of Python per quantizer call, forward and backward (MobileBERT W4A4: 1333 quantizer calls per
    forward); the replay pays none.

    * `optimizer` must be capture-safe: torch.optim.SGD, or Adam / AdamW constructed with ``capturable=True``;
    * inputs / targets are copied into static buffers of the examples' shapes; the returned loss is the graph's static
      output (clone it to keep a value across calls);
    * the warm-up iterations (allocator pools, workspaces, optimizer state) are real optimisation steps on the example
      batch; with `restore_state=True` (default) parameters, buffers and optimizer state are put back to their values
      from before the warm-up after capture, so training starts from the model that was passed in;
    * estimator state must not change during the step: ranges fixed (`fix_ranges`) or learnable, not estimating;
    * `grad_sync`: a `quantization.data_parallel.GradientBuckets` over the module's parameters makes this the
      DATA-PARALLEL step (BASELINE configs[4]): the bucketed gradient all-reduces -- raw RCCL, on a second stream that
      forks from and joins the capturing stream -- are part of the recorded graph, overlapped with the rest of backward.
    """

    def __init__(self, module, loss_fn, optimizer, example_inputs, example_targets=(), warmup=3, restore_state=True,
                 grad_sync=None):
        tensors = tuple(example_inputs) + tuple(example_targets)
        if not tensors or not all(torch.is_tensor(t) and t.is_cuda for t in tensors):
            raise ValueError('GraphedTrainStep needs ROCm tensors as example inputs / targets')
        _refuse_c10d_exchange('GraphedTrainStep')
        self.module, self.loss_fn, self.optimizer, self.grad_sync = module, loss_fn, optimizer, grad_sync
        self.static_inputs = tuple(t.clone() for t in example_inputs)
        self.static_targets = tuple(t.clone() for t in example_targets)
        snap = {k: v.clone() for k, v in module.state_dict().items()} if restore_state else None
        # optimizer state as it was handed in (empty for a fresh optimizer; moments + step counters of one that has
        # already stepped): restored IN PLACE after capture -- the graph holds the addresses of the live tensors
        opt_snap = None
        if restore_state:
            opt_snap = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                        for p, st in optimizer.state.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        _quiesce()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
            self.static_loss = self._step()
        self._cache_refs = derived_cache_tensors(module)       # read by address in every replay: see derived_cache_tensors
        if snap is not None:
            live = module.state_dict()
            for k, v in live.items():
                if k not in snap or v.shape != snap[k].shape:
                    raise RuntimeError(f'module state {k} changed shape during capture')
                v.copy_(snap[k])
            for p, st in optimizer.state.items():
                before = opt_snap.get(id(p), {})
                for k, v in st.items():
                    if torch.is_tensor(v):
                        if torch.is_tensor(before.get(k)) and before[k].shape == v.shape:
                            v.copy_(before[k])
                        else:
                            v.zero_()            # state created by the warm-up: back to "no step taken"
        options.invalidate_derived_caches()

    def _step(self):
        if self.grad_sync is not None:
            self.grad_sync.zero_()               # gradients stay views of the flat buckets
        else:
            self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn(self.module(*self.static_inputs), *self.static_targets)
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync.finish()
        self.optimizer.step()
        return loss.detach()

    def __call__(self, inputs, targets=()):
        for dst, src in zip(self.static_inputs + self.static_targets, tuple(inputs) + tuple(targets)):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError(f'input shape / dtype changed: captured {tuple(dst.shape)} {dst.dtype}, '
                                 f'got {tuple(src.shape)} {src.dtype}')
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        # the replay rewrote weights / learnable ranges in place without bumping tensor._version: derived caches
        # (int8 weight indices, NoNorm parameters, stacked operands) must not survive it
        options.invalidate_derived_caches()
        return self.static_loss
