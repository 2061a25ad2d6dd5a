"""Model-level AdaRound driver (API counterpart of the reference's utils/adaround_utils.py: ``get_train_samples``
:19-32, ``apply_adaround_to_model`` :35-139).

Layers are optimised strictly one after the other in module order: with ``asym`` (the default) layer k is trained on
the inputs it sees AFTER the rounding of layers < k was learned, so there is no layer parallelism to exploit.  What
does shard is the data: with ``quantization.distributed`` enabled every rank keeps the cached layer I/O of the samples
``i % world == rank`` and the per-step weight gradient is SUM all-reduced over RCCL before the fused Adam update
(quantization/adaround/adaround.py), so all ranks walk through the same sequence of roundings in lock-step.
Sample collection stays replicated: every rank reads the same `data_loader` (the samples are a few MB of token ids).

With ``asym=False`` the per-layer problems are independent (inputs AND targets come from the FP32 network), so the
LAYERS shard instead: rank r optimises layers r, r + world, ... on the full sample set with no collective in the data
path, and every layer's learned state (alpha, the range buffers a grid init may have moved, its losses) is broadcast
from its owner at the end -- one small exchange per layer instead of a [N, K] gradient all-reduce per iteration.
(`config.adaround.layer_parallel = False` keeps the data-parallel scheme.)
"""
import logging

import torch

from quantization import distributed as tq_dist
from quantization.adaround import apply_adaround_to_layer
from quantization.adaround.adaround import install_adaround_quantizer
from quantization.adaround.utils import AdaRoundActQuantMode, AdaRoundInitMode
from quantization.base_quantized_classes import QuantizedModule
from utils.utils import pass_data_for_range_estimation, Stopwatch

logger = logging.getLogger('AdaRound')
logger.setLevel(logging.INFO)

_SUPPORTED_ACT_MODES = (AdaRoundActQuantMode.no_act_quant, AdaRoundActQuantMode.post_adaround)


def get_train_samples(data_loader, num_samples, return_labels=False, inp_idx=0, lbl_idx=1):
    """The first `num_samples` inputs (and labels) of `data_loader`, concatenated along dim 0."""
    inputs, labels, have = [], [], 0
    for batch in data_loader:
        inputs.append(batch[inp_idx])
        if return_labels:
            labels.append(batch[lbl_idx])
        have += batch[inp_idx].size(0)
        if have >= num_samples:
            break
    x = torch.cat(inputs, dim=0)[:num_samples]
    return (x, torch.cat(labels, dim=0)[:num_samples]) if return_labels else x


def adaround_layers(model, wanted):
    """[(name, module)] of the QuantizedModules with a weight that `wanted` ('all' or explicit names) selects,
    in the order model.named_modules() yields them (= the order they are optimised in)."""
    owners = [(n, m) for n, m in model.named_modules() if isinstance(m, QuantizedModule) and hasattr(m, 'weight')]
    if 'all' in wanted:
        return owners
    known = dict(owners)
    for name in wanted:
        if name not in known:
            logger.warning(f'skipping unknown layer {name}')
    return [(n, m) for n, m in owners if n in wanted]


def _gather_layer_results(todo, results, cfg, world, rank):
    """Layer-parallel AdaRound: every layer's learned rounding (alpha), its range buffers and its losses go from the
    owner (layer index % world) to everybody.  Non-owners first build the same AdaRound quantizer module."""
    dist, grp = torch.distributed, tq_dist.group()
    for i, (name, module) in enumerate(todo):
        owner = i % world
        src = dist.get_global_rank(grp, owner) if grp is not None else owner
        if rank != owner:
            q = install_adaround_quantizer(module, cfg)
            with torch.no_grad():
                q(module.weight)                      # allocates alpha with the right shape (values replaced below)
            q.soft_targets = False
            module.caching = True
        q = module.weight_quantizer.quantizer
        for t in (q.alpha.data, q._delta, getattr(q, '_zero_float', None)):
            if t is not None:
                dist.broadcast(t, src=src, group=grp)
        sgn = getattr(q, '_signed', None)
        if sgn is not None:
            flag = sgn.to(torch.uint8).reshape(1).clone()
            dist.broadcast(flag, src=src, group=grp)
            q._signed = flag.reshape(()).to(torch.bool)
        box = [results.get(name)]
        dist.broadcast_object_list(box, src=src, group=grp)
        results[name] = box[0]


def apply_adaround_to_model(config, model, data_loader, range_est_data_loader, batch_size, driver=None,
                            get_samples_fn=get_train_samples, inp_idx=0):
    """AdaRound on every selected layer of a QuantizedModel, then (act mode `post_adaround`) a fresh activation
    calibration on top of the learned weights.  Returns {layer name: DotDict of local losses}."""
    cfg = config.adaround
    if cfg.act_quant_mode not in _SUPPORTED_ACT_MODES:
        raise NotImplementedError(f"act mode '{cfg.act_quant_mode}' is not implemented")

    device = next(model.parameters()).device
    samples = get_samples_fn(data_loader, num_samples=cfg.num_samples).to(device)
    todo = adaround_layers(model, cfg.layers)
    if not todo:
        logger.warning('No layers to apply AdaRound for, exiting...')
        return {}
    if tq_dist.is_enabled():
        logger.info(f'data-parallel AdaRound: {samples.size(0)} samples over '
                    f'{torch.distributed.get_world_size()} ranks')

    # AdaRound itself always runs against FP32 activations
    config.quant.act_quant = False
    model.reset_act_ranges()
    model.full_precision_acts()

    layer_parallel = (tq_dist.is_enabled() and not cfg.asym and cfg.get('layer_parallel', True)
                      and cfg.init != AdaRoundInitMode.mse_out_asym)
    world = torch.distributed.get_world_size(tq_dist.group()) if layer_parallel else 1
    rank = torch.distributed.get_rank(tq_dist.group()) if layer_parallel else 0
    if layer_parallel:
        logger.info(f'layer-parallel AdaRound: {len(todo)} independent layers over {world} ranks')

    results, total = {}, Stopwatch()
    for i, (name, module) in enumerate(todo):
        if i % world != rank:
            continue                                  # another rank's layer; its result arrives below
        logger.info(f'Started AdaRound for layer {name}')
        model.full_precision()
        module.quantized_weights()
        total.start()
        with Stopwatch() as per_layer:
            if layer_parallel:
                with tq_dist.suspended():             # a complete local problem: nothing to exchange inside
                    results[name] = apply_adaround_to_layer(model, module, samples, batch_size=batch_size,
                                                            act_quant=config.quant.act_quant, adaround_config=cfg)
            else:
                results[name] = apply_adaround_to_layer(model, module, samples, batch_size=batch_size,
                                                        act_quant=config.quant.act_quant, adaround_config=cfg)
        total.stop()
        logger.info(f'Done AdaRound for layer {name}. {per_layer.format()}\n')
    logger.info(f'Done optimizing all layers. {total.format()}')
    if layer_parallel:
        _gather_layer_results(todo, results, cfg, world, rank)

    if cfg.act_quant_mode == AdaRoundActQuantMode.post_adaround:
        if driver is not None:
            model.quantized_weights()
            state = driver.validate()
            logger.info(f"FINAL res (without acts quant):\t{state.metrics['top_1_accuracy'] * 100:.2f}%")
        config.quant.act_quant = True
        model.estimate_act_ranges()
        pass_data_for_range_estimation(loader=range_est_data_loader, model=model, act_quant=True, weight_quant=True,
                                       max_num_batches=config.act_quant.num_batches,
                                       cross_entropy_layer=config.act_quant.cross_entropy_layer, inp_idx=inp_idx)
        model.fix_act_ranges()

    model.set_quant_state(weight_quant=True, act_quant=config.quant.act_quant)
    return results
