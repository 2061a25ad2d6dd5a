"""Model-level AdaRound driver (counterpart of the reference's utils/adaround_utils.py :19-139):
collect samples, walk the QuantizedModules that own a weight, optimise them one after the other
(layer k sees the rounding of layers < k), then re-calibrate the activation quantizers."""
import logging

import torch

from quantization.adaround import apply_adaround_to_layer
from quantization.adaround.utils import AdaRoundActQuantMode
from quantization.base_quantized_classes import QuantizedModule
from utils.utils import pass_data_for_range_estimation, Stopwatch

logger = logging.getLogger('AdaRound')
logger.setLevel(logging.INFO)


def get_train_samples(data_loader, num_samples, return_labels=False, inp_idx=0, lbl_idx=1):
    xs, ys, seen = [], [], 0
    for batch in data_loader:
        xs.append(batch[inp_idx])
        if return_labels:
            ys.append(batch[lbl_idx])
        seen += batch[inp_idx].size(0)
        if seen >= num_samples:
            break
    x = torch.cat(xs, dim=0)[:num_samples]
    if return_labels:
        return x, torch.cat(ys, dim=0)[:num_samples]
    return x


def _select_layers(model, wanted):
    candidates = [name for name, m in model.named_modules()
                  if isinstance(m, QuantizedModule) and hasattr(m, 'weight')]
    if 'all' in wanted:
        return candidates
    chosen = []
    for name in wanted:
        if name in candidates:
            chosen.append(name)
        else:
            logger.warning(f'skipping unknown layer {name}')
    return chosen


def apply_adaround_to_model(config, model, data_loader, range_est_data_loader, batch_size,
                            driver=None, get_samples_fn=get_train_samples, inp_idx=0):
    """Apply AdaRound to every selected layer of `model` (a QuantizedModel), in module order."""
    train_data = get_samples_fn(data_loader, num_samples=config.adaround.num_samples)
    train_data = train_data.to(next(model.parameters()).device)

    layer_names = _select_layers(model, config.adaround.layers)
    if not layer_names:
        logger.warning('No layers to apply AdaRound for, exiting...')
        return

    if config.adaround.act_quant_mode not in (AdaRoundActQuantMode.no_act_quant,
                                              AdaRoundActQuantMode.post_adaround):
        raise NotImplementedError(
            f"act mode '{config.adaround.act_quant_mode}' is not implemented")
    config.quant.act_quant = False
    model.reset_act_ranges()
    model.full_precision_acts()

    total = Stopwatch()
    for name, module in model.named_modules():
        if name not in layer_names:
            continue
        logger.info(f'Started AdaRound for layer {name}')
        model.full_precision()
        module.quantized_weights()
        total.start()
        with Stopwatch() as per_layer:
            apply_adaround_to_layer(model, module, train_data, batch_size=batch_size,
                                    act_quant=config.quant.act_quant,
                                    adaround_config=config.adaround)
        logger.info(f'Done AdaRound for layer {name}. {per_layer.format()}\n')
        total.stop()
    logger.info(f'Done optimizing all layers. {total.format()}')

    if config.adaround.act_quant_mode == AdaRoundActQuantMode.post_adaround:
        if driver is not None:
            model.quantized_weights()
            state = driver.validate()
            logger.info('FINAL res (without acts quant):\t'
                        f"{state.metrics['top_1_accuracy'] * 100:.2f}%")
        config.quant.act_quant = True
        model.estimate_act_ranges()
        pass_data_for_range_estimation(
            loader=range_est_data_loader, model=model, act_quant=True, weight_quant=True,
            max_num_batches=config.act_quant.num_batches,
            cross_entropy_layer=config.act_quant.cross_entropy_layer, inp_idx=inp_idx)
        model.fix_act_ranges()

    model.set_quant_state(weight_quant=True, act_quant=config.quant.act_quant)
