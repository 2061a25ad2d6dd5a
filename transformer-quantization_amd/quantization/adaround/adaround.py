"""Per-layer AdaRound optimisation (counterpart of the reference's
quantization/adaround/adaround.py: ``apply_adaround_to_layer`` :27-136, ``apply_mse_init`` :160-178,
``apply_mse_out_init`` :181-201, ``optimize_local_loss`` :204-260).

One optimisation step on the MI355X path:

    K10  tq_adaround_fwd        w_q = s * (clamp(floor(w/s) + h(alpha) (+zp)) - zp)
    K12  layer.run_forward      GEMM forward/backward through torch (hipBLASLt)
    K13  reconstruction loss    mse(out, target, 'none').sum(1).mean()
    K11  tq_adaround_bwd_adam   d loss/d alpha + regulariser gradient + Adam update, in place

instead of ~25 element-wise ATen kernels + a generic optimizer.  With
``quantization.distributed`` enabled the cached samples are sharded over the ranks
(sample i lives on rank i % world) and the weight gradient is SUM all-reduced before K11, which
reproduces the single-rank step on the same global batch.
"""
import logging
from math import ceil

import numpy as np
import torch
import torch.nn.functional as F

from quantization import _hip
from quantization import distributed as tq_dist
from quantization.adaround.quantizer import ADAROUND_QUANTIZER_MAP
from quantization.adaround.utils import (
    MODE_TO_LOSS_TYPE,
    AdaRoundInitMode,
    AdaRoundLossType,
    CombinedLoss,
    GetLayerInpOut,
    LayerOutputMSE,
)
from quantization.range_estimators import candidate_params
from utils.utils import DotDict

logger = logging.getLogger('AdaRound')
logger.setLevel(logging.INFO)


class FusedAlphaAdam:
    """Adam state (exp_avg, exp_avg_sq, step) for the rounding variable; ``step`` is K11.
    Hyper-parameters follow torch.optim.Adam's defaults, which is what the reference uses
    (adaround/adaround.py:98-99)."""

    def __init__(self, quantizer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.quantizer = quantizer
        self.lr, self.betas, self.eps = lr, betas, eps
        alpha = quantizer.alpha
        self.exp_avg = torch.zeros_like(alpha)
        self.exp_avg_sq = torch.zeros_like(alpha)
        self.t = 0

    def zero_grad(self):
        pass

    def step(self, w, grad_wq, reg_weight, beta, want_grad=False):
        self.t += 1
        q = self.quantizer
        return _hip.backend().adaround_bwd_adam(
            w, grad_wq, q.alpha.data, self.exp_avg, self.exp_avg_sq, q.kernel_args(w),
            q.mode_code(), q.temperature, reg_weight, beta, self.lr, self.betas[0], self.betas[1],
            self.eps, self.t, want_grad=want_grad)


    # ---- graph replay support ---------------------------------------------------------------------------
    def schedule_row(self, t, reg_weight, beta):
        """(reg_weight, beta, 1 - b1^t, sqrt(1 - b2^t)) of Adam step t as the fp32 values `step` hands to K11: the
        betas are narrowed to fp32 first (they cross the C ABI as floats), the powers are evaluated in double."""
        import math
        import numpy as np
        b1, b2 = float(np.float32(self.betas[0])), float(np.float32(self.betas[1]))
        return (np.float32(reg_weight), np.float32(beta), np.float32(1.0 - math.pow(b1, t)),
                np.float32(math.sqrt(1.0 - math.pow(b2, t))))

    def step_scheduled(self, w, grad_wq, sched_row):
        """`step` with its per-iteration scalars in the device tensor `sched_row` (fp32 [4], see schedule_row)."""
        self.t += 1
        q = self.quantizer
        _hip.backend().adaround_bwd_adam_sched(w, grad_wq, q.alpha.data, self.exp_avg, self.exp_avg_sq, q.kernel_args(w),
                                               q.mode_code(), q.temperature, sched_row, self.lr, self.betas[0],
                                               self.betas[1], self.eps)


class _GraphedIterations:
    """Iterations first..iters of the fused AdaRound loop as replays of ONE hipGraph.

    The body is recorded (not executed) after the eager warm-up iterations; what differs between iterations -- the
    sample indices and K11's four scalars -- comes from device tables indexed by a device-side iteration counter, so a
    replay needs no host input at all."""

    def __init__(self, layer, q, optimizer, loss_fn, cached_inps, cached_outs, idx_rows, first, iters):
        import numpy as np
        be = _hip.backend()
        dev = cached_inps.device
        self.layer, self.q, self.opt = layer, q, optimizer
        self.idx_all = torch.stack(idx_rows).to(dev)                               # [iters - first + 1, batch]
        rows = [optimizer.schedule_row(it, loss_fn.weight if loss_fn.schedule(it)[1] else 0.0, loss_fn.schedule(it)[0])
                for it in range(first, iters + 1)]
        self.sched_all = torch.from_numpy(np.asarray(rows, dtype=np.float32)).to(dev)
        self.pos = torch.zeros(1, dtype=torch.int64, device=dev)
        w = layer.weight
        bias = layer.bias if hasattr(layer, 'bias') else None

        def body():
            idx = self.idx_all.index_select(0, self.pos).view(-1)
            sched = self.sched_all.index_select(0, self.pos).view(-1)
            cur_inp = cached_inps.index_select(0, idx)
            cur_out = cached_outs.index_select(0, idx)
            w_q = be.adaround_fwd(w, q.alpha.detach(), q.kernel_args(w), q.mode_code(), True, q.temperature)
            with torch.no_grad():               # closed-form gradient of the plain Linear, as in the eager loop
                out = layer.run_forward(cur_inp, w_q.detach(), bias)
                n_means = out.numel() // out.size(1) if out.dim() > 1 else 1
                grad_out = (out - cur_out) * (2.0 / n_means)
                grad_wq = grad_out.reshape(-1, grad_out.shape[-1]).t().mm(cur_inp.reshape(-1, cur_inp.shape[-1]))
            optimizer.step_scheduled(w, grad_wq, sched)
            self.pos.add_(1)
            return out.detach(), cur_out

        t_before = optimizer.t
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out, self.cur_out = body()
        optimizer.t = t_before                        # recording executed nothing

    def replay(self):
        self.opt.t += 1
        self.graph.replay()


def install_adaround_quantizer(layer, cfg):
    """Swap in the AdaRound flavour of `layer`'s weight quantizer, sharing the range buffers (reference
    adaround/adaround.py:75-90).  Separate from the optimisation so that a rank that did not optimise this layer
    (layer-parallel AdaRound, utils/adaround_utils.py) can build the same module and receive the learned state."""
    org_q = layer.weight_quantizer.quantizer
    if org_q.__class__ not in ADAROUND_QUANTIZER_MAP:
        raise NotImplementedError(f'AdaRound is not supported for "{org_q.__class__}"')
    w_quantizer = ADAROUND_QUANTIZER_MAP[org_q.__class__](
        n_bits=org_q.n_bits, scale_domain=org_q.scale_domain, per_channel=org_q.per_channel,
        eps=org_q.eps)
    for name in ('_delta', '_zero_float', '_signed'):
        if hasattr(org_q, name):
            w_quantizer.register_buffer(name, getattr(org_q, name))
    layer.weight_quantizer.quantizer = w_quantizer
    w_quantizer.round_mode = cfg.round_mode
    w_quantizer.temperature = cfg.annealing[0]
    return w_quantizer


def apply_adaround_to_layer(model, layer, data_tensor, batch_size, act_quant, adaround_config,
                            keep_gpu=True):
    """Learn the rounding of `layer`'s weights so that its output matches the FP32 layer."""
    cfg = adaround_config
    layer.caching = False

    # weight grid
    if cfg.init == AdaRoundInitMode.range_estimator:
        pass
    elif cfg.init == AdaRoundInitMode.mse:
        apply_mse_init(layer)
    elif cfg.init == AdaRoundInitMode.mse_out:
        apply_mse_out_init(model, layer, data_tensor, batch_size)
    elif cfg.init == AdaRoundInitMode.mse_out_asym:
        apply_mse_out_init(model, layer, data_tensor, batch_size, asym=True)
    else:
        raise ValueError(f'Unknown initialization for AdaRound: {cfg.init}')

    if not cfg.include_act_func:
        org_act_func = layer.activation_function
        layer.activation_function = None

    w_quantizer = install_adaround_quantizer(layer, cfg)

    # first pass also initialises alpha
    get_inp_out = GetLayerInpOut(model, layer, asym=cfg.asym, act_quant=act_quant)
    inp, out = get_inp_out(data_tensor[:batch_size])
    loss_soft_before, loss_hard_before = _compute_and_display_local_losses(
        w_quantizer, layer, inp, out, infix='before optimization')
    w_quantizer.soft_targets = True

    loss_fn = CombinedLoss(
        quantizer=w_quantizer, loss_type=MODE_TO_LOSS_TYPE[w_quantizer.round_mode],
        weight=cfg.weight, max_count=cfg.iters, b_range=cfg.annealing, warmup=cfg.warmup,
        decay_type=cfg.decay_type, decay_shape=cfg.decay_shape, decay_start=cfg.decay_start)
    optimizer = FusedAlphaAdam(w_quantizer, lr=cfg.lr)

    optimize_local_loss(layer, get_inp_out, data_tensor, optimizer, loss_fn, batch_size, cfg.iters,
                        keep_gpu=keep_gpu)

    logger.info(f'Local loss before optimization (hard quant): {loss_hard_before:.7f}')
    loss_soft_after, loss_hard_after = _compute_and_display_local_losses(
        w_quantizer, layer, inp, out, infix='after optimization')

    w_quantizer.soft_targets = False   # hard up/down decision from now on
    if not cfg.include_act_func:
        layer.activation_function = org_act_func
    layer.caching = True

    return DotDict(loss_soft_before=loss_soft_before, loss_hard_before=loss_hard_before,
                   loss_soft_after=loss_soft_after, loss_hard_after=loss_hard_after)


def _compute_and_display_local_losses(quantizer, layer, inp, out, infix=''):
    keep = quantizer.soft_targets
    losses = []
    with torch.no_grad():
        for soft in (True, False):
            quantizer.soft_targets = soft
            losses.append(float(F.mse_loss(layer(inp), out)))
    quantizer.soft_targets = keep
    if infix:
        infix = infix.strip() + ' '
    logger.info(f'Local loss {infix}(soft quant): {losses[0]:.7f}')
    logger.info(f'Local loss {infix}(hard quant): {losses[1]:.7f}')
    return losses[0], losses[1]


def _shrink_candidates(w):
    """The 80 symmetric ranges s_i = |w|max * (1 - 0.01 i) of the reference's MSE init, as fp32."""
    mn, mx = _hip.backend().minmax(w.detach(), 1, 1)
    w_absmax = torch.max(mx, torch.abs(mn))
    factors = torch.tensor([1.0 - 0.01 * i for i in range(80)], dtype=torch.float64).float()
    return w_absmax, (w_absmax.cpu() * factors).numpy()


def apply_mse_init(layer):
    """Pick the symmetric range that minimises the weight quantization MSE: all 80 candidates in
    one pass of ``tq_mse_candidates`` (the reference loops set_quant_range + quantize + mse_loss,
    adaround/adaround.py:160-178)."""
    w = layer.weight
    q = layer.weight_quantizer.quantizer
    be = _hip.backend()
    with torch.no_grad():
        _, s = _shrink_candidates(w)
        table = candidate_params(-s, s, q.n_bits, q.symmetric, q.eps)
        cand = be.candidate_table(table, w.device)
        loss = be.zeros_f64((1, len(s)), w.device)
        be.mse_candidates(w, 1, cand, loss)
        thr = be.candidate_table(np.stack([-s, s]).astype(np.float32), w.device)
        xmin, xmax, best = be.argmin_select(loss, thr[0], thr[1])
        best_max = xmax[0]
        logger.info(f'Finished: set max={float(best_max):.3f} '
                    f'(mse={float(loss[0, int(best[0])]) / w.numel():.7f})')
        q.set_quant_range(-best_max, best_max)


def apply_mse_out_init(model, layer, data_tensor, batch_size, asym=False):
    """The same 80 candidate grids, scored on the layer OUTPUT error (reference adaround/adaround.py:181-201).  The
    80 scores stay on the device (LayerOutputMSE adds fp64 cells through tq_recon_loss); first-minimum selection and
    the final set_quant_range run there too: one host synchronisation per layer (for the log line) instead of the
    reference's 80 x #mini-batches `.item()` calls."""
    w = layer.weight
    q = layer.weight_quantizer.quantizer
    get_inp_out = GetLayerInpOut(model, layer, asym=asym)
    loss_fn = LayerOutputMSE(layer, get_inp_out, data_tensor, batch_size)
    with torch.no_grad():
        w_absmax, _ = _shrink_candidates(w)
        cands = [w_absmax * (1.0 - 0.01 * i) for i in range(80)]
        scores = []
        for s in cands:
            q.set_quant_range(-s, s)
            scores.append(loss_fn())
        scores = torch.stack(scores)
        best = torch.argmin(scores)                       # first minimum, like the reference's strict `<` scan
        best_max = torch.stack(cands)[best]
        logger.info(f'Finished: set max={float(best_max):.3f} (mse={float(scores[best]):.7f})')
        q.set_quant_range(-best_max, best_max)
    return scores


def _fused_step_is_exact(layer):
    """The closed-form / fused AdaRound step optimises run_forward(+activation function) against the cached FP32
    output; that equals the reference's objective iff nothing else happens in layer.forward."""
    from quantization.quantization_manager import Qstates
    if getattr(layer, '_quant_a', False):
        return False
    mgr = layer.weight_quantizer
    return getattr(mgr, 'state', Qstates.fix_ranges) == Qstates.fix_ranges


def _cache_layer_io(get_inp_out, data_tensor, batch_size, keep_gpu):
    """Layer inputs / FP32 outputs for every sample this rank owns, resident in HBM when they
    fit (288 GB per MI355X: BERT-base needs 0.4 + 1.6 GB for the widest layer)."""
    ws = torch.distributed.get_world_size() if tq_dist.is_enabled() else 1
    rk = torch.distributed.get_rank() if tq_dist.is_enabled() else 0
    own = data_tensor[rk::ws] if ws > 1 else data_tensor
    inps, outs = [], []
    with torch.no_grad():
        for i in range(ceil(own.size(0) / batch_size)):
            cur_inp, cur_out = get_inp_out(own[i * batch_size:(i + 1) * batch_size])
            inps.append(cur_inp if keep_gpu else cur_inp.cpu())
            outs.append(cur_out if keep_gpu else cur_out.cpu())
        device = cur_inp.device
    return torch.cat(inps), torch.cat(outs), device, ws, rk


def optimize_local_loss(layer, get_inp_out, data_tensor, optimizer, loss_fn, batch_size, iters,
                        use_cached_data=True, keep_gpu=True, batch_indices=None, on_step=None):
    """AdaRound optimisation loop.

    `optimizer` is a ``FusedAlphaAdam`` (fused K11 path, default) or any torch optimizer over
    ``quantizer.alpha`` (generic autograd path through ``_AdaRoundFn`` and ``loss_fn``).
    `batch_indices` optionally fixes the sample indices of every iteration (tests); otherwise
    ``torch.randperm`` on the global RNG is used like the reference (:236).  `on_step(it, layer)` is called
    after every update (trace replay in the parity tests)."""
    fused = isinstance(optimizer, FusedAlphaAdam)
    if fused and not _fused_step_is_exact(layer):
        # The fused step evaluates run_forward (+ activation function) only.  With the layer's ACTIVATION quantizer on,
        # or the weight manager still estimating ranges, the reference's objective loss_fn(layer(x), target)
        # (adaround/adaround.py:246-248) contains more than that: take the generic autograd path through layer.forward.
        if tq_dist.is_enabled():
            raise NotImplementedError('data-parallel AdaRound needs the fused step: switch the layer\'s activation '
                                      'quantizer off (model.full_precision(); layer.quantized_weights()) and fix the '
                                      'weight ranges first, as apply_adaround_to_model does')
        logger.info('activation quantizer on / weight ranges not fixed: generic autograd AdaRound step')
        q_ = layer.weight_quantizer.quantizer
        optimizer = torch.optim.Adam([q_.alpha], lr=optimizer.lr, betas=optimizer.betas, eps=optimizer.eps)
        fused = False
    if use_cached_data:
        logger.info('Caching data for local loss optimization')
        cached_inps, cached_outs, device, ws, rk = _cache_layer_io(
            get_inp_out, data_tensor, batch_size, keep_gpu)
        # The sampling population is the set of CACHED rows, as upstream (:236): for most layers one row per
        # sample, but e.g. BERT's position embeddings see one [1, T] input per forward, i.e. one row per
        # cached batch.  Sharded: row i lives on rank i % ws; ranks may differ by one row, so use the minimum.
        rows = cached_inps.size(0)
        if ws > 1:
            r = torch.tensor([rows], device=device, dtype=torch.int64)
            torch.distributed.all_reduce(r, op=torch.distributed.ReduceOp.MIN)
            rows = int(r)
        n_total = rows * ws
    else:
        ws, rk, device = 1, 0, layer.weight.device
        n_total = data_tensor.size(0)
    q = layer.weight_quantizer.quantizer
    be = _hip.backend()
    # plain Linear layers without a folded activation function: closed-form gradient, no autograd
    from quantization.autoquant_utils import QuantLinear
    manual_linear = fused and type(layer) is QuantLinear and layer.activation_function is None

    # hipGraph replay of the iterations after a short eager warm-up (options.GRAPH_ADAROUND): same kernels, same order,
    # same sample sequence -- the indices of ALL iterations are drawn up front, in the order the eager loop draws them
    from quantization import options
    n_warm = 3
    graphable = (fused and manual_linear and options.GRAPH_ADAROUND and use_cached_data and ws == 1 and on_step is None
                 and cached_inps.is_cuda and iters > n_warm + 8 and n_total >= 1
                 and loss_fn.loss_type == AdaRoundLossType.relaxation and hasattr(be, 'adaround_bwd_adam_sched'))
    drawn = None
    if graphable:
        drawn = [(torch.as_tensor(batch_indices[i]) if batch_indices is not None
                  else torch.randperm(n_total)[:batch_size]) for i in range(iters)]
        if len({d.numel() for d in drawn}) != 1 or max(int(d.max()) for d in drawn) >= cached_inps.size(0):
            graphable = False                  # ragged / out-of-range index rows: the eager loop reports them
    replayer = None

    for i in range(iters):
        if graphable and i == n_warm and replayer is None:
            try:
                replayer = _GraphedIterations(layer, q, optimizer, loss_fn, cached_inps, cached_outs, drawn[n_warm:],
                                              n_warm + 1, iters)
            except Exception as e:       # noqa: BLE001 -- recording executes nothing: carry on eagerly
                logger.info(f'AdaRound loop not captured as a hipGraph ({e!r}); eager iterations')
                graphable = False
                torch.cuda.synchronize()
        if replayer is not None:
            it = i + 1
            loss_fn.iter = it
            replayer.replay()
            if it == 1 or it % 100 == 0:
                b, reg_on = loss_fn.schedule(it)
                round_loss = float(be.adaround_reg(q.alpha, q.mode_code(), q.temperature, b, loss_fn.weight)) if reg_on else 0.0
                rec_v = float(be.recon_loss(replayer.out, replayer.cur_out))
                logger.info(f'Total loss:\t{rec_v + round_loss:.4f} (rec:{rec_v:.4f}, '
                            f'round:{round_loss:.3f})\tb={b:.2f}\titer={it}')
            continue
        idx = (drawn[i] if drawn is not None else
               (torch.as_tensor(batch_indices[i]) if batch_indices is not None
                else torch.randperm(n_total)[:batch_size]))
        n_global = idx.numel()
        if use_cached_data:
            if ws > 1:
                idx = idx[idx % ws == rk] // ws      # samples of this rank, local positions
            if idx.numel() and int(idx.max()) >= cached_inps.size(0):
                raise IndexError(f'sample index {int(idx.max())} outside the {cached_inps.size(0)} cached rows')
            cur_inp = cached_inps[idx].to(device)
            cur_out = cached_outs[idx].to(device)
        else:
            cur_inp, cur_out = get_inp_out(data_tensor[idx])

        if not fused:
            optimizer.zero_grad()
            loss = loss_fn(layer(cur_inp), cur_out)
            loss.backward()
            optimizer.step()
            if on_step is not None:
                on_step(i + 1, layer)
            continue

        # ---- fused path -----------------------------------------------------------------
        it = i + 1
        loss_fn.iter = it
        b, reg_on = loss_fn.schedule(it)
        w = layer.weight
        w_q = be.adaround_fwd(w, q.alpha.detach(), q.kernel_args(w), q.mode_code(), True,
                              q.temperature)
        if not manual_linear:
            w_q.requires_grad_(True)
        bias = layer.bias if hasattr(layer, 'bias') else None
        if cur_inp.size(0) == 0:
            rec = torch.zeros((), device=device)
            grad_wq = torch.zeros_like(w_q)
        elif manual_linear:
            # no autograd: out = x W_q^T + b ; dL/dout = 2 (out - tgt) * share / (#means) ;
            # dL/dW_q = dL/dout^T x   (two hipBLASLt GEMMs + one element-wise kernel)
            with torch.no_grad():
                out = layer.run_forward(cur_inp, w_q.detach(), bias)
                share = cur_inp.size(0) / n_global
                n_means = out.numel() // out.size(1) if out.dim() > 1 else 1
                diff = out - cur_out
                grad_out = diff * (2.0 * share / n_means)
                grad_wq = grad_out.reshape(-1, grad_out.shape[-1]).t().mm(
                    cur_inp.reshape(-1, cur_inp.shape[-1]))
                rec = None      # evaluated only when it is logged
        else:
            out = layer.run_forward(cur_inp, w_q, bias)
            if layer.activation_function is not None:
                out = layer.activation_function(out)
            # local share of the global-batch mean: sum(1).mean() * n_local / n_global
            rec = F.mse_loss(out, cur_out, reduction='none').sum(1).mean() * (
                cur_inp.size(0) / n_global)
            rec.backward()
            grad_wq = w_q.grad
        grad_wq = tq_dist.sync_sum(grad_wq)
        optimizer.step(w, grad_wq, loss_fn.weight if reg_on else 0.0, b)
        if loss_fn.loss_type == AdaRoundLossType.temp_decay and it >= loss_fn.loss_start:
            # like CombinedLoss.__call__ (reference adaround/utils.py:154-157): the annealed
            # temperature takes effect from the NEXT forward on
            q.temperature = b

        if on_step is not None:
            on_step(it, layer)
        if it == 1 or it % 100 == 0:
            round_loss = float(be.adaround_reg(q.alpha, q.mode_code(), q.temperature, b,
                                               loss_fn.weight)) if reg_on else 0.0
            if rec is None:
                rec = be.recon_loss(out, cur_out) * (cur_inp.size(0) / n_global)
            rec_v = float(tq_dist.sync_sum(rec.detach().clone().double()))
            logger.info(f'Total loss:\t{rec_v + round_loss:.4f} (rec:{rec_v:.4f}, '
                        f'round:{round_loss:.3f})\tb={b:.2f}\titer={it}')
