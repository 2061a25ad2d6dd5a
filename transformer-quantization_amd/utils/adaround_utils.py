"""Model-level AdaRound driver (API counterpart of the reference's utils/adaround_utils.py: ``get_train_samples``
:19-32, ``apply_adaround_to_model`` :35-139).

Layers are optimised strictly one after the other in module order: with ``asym`` (the default) layer k is trained on
the inputs it sees AFTER the rounding of layers < k was learned, so there is no layer parallelism to exploit.  What
does shard is the data: with ``quantization.distributed`` enabled every rank keeps the cached layer I/O of the samples
``i % world == rank`` and the per-step weight gradient is SUM all-reduced over RCCL before the fused Adam update
(quantization/adaround/adaround.py), so all ranks walk through the same sequence of roundings in lock-step.
Sample collection stays replicated: every rank reads the same `data_loader` (the samples are a few MB of token ids).
"""
import logging

import torch

from quantization import distributed as tq_dist
from quantization.adaround import apply_adaround_to_layer
from quantization.adaround.utils import AdaRoundActQuantMode
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

    results, total = {}, Stopwatch()
    for name, module in todo:
        logger.info(f'Started AdaRound for layer {name}')
        model.full_precision()
        module.quantized_weights()
        total.start()
        with Stopwatch() as per_layer:
            results[name] = apply_adaround_to_layer(model, module, samples, batch_size=batch_size,
                                                    act_quant=config.quant.act_quant, adaround_config=cfg)
        total.stop()
        logger.info(f'Done AdaRound for layer {name}. {per_layer.format()}\n')
    logger.info(f'Done optimizing all layers. {total.format()}')

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
