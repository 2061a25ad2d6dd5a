"""QAT preparation (counterpart of the reference's utils/qat_utils.py :14-45): calibrate, then
either make the ranges learnable or keep estimating them during training."""
import logging

from utils.utils import pass_data_for_range_estimation

logger = logging.getLogger('QAT')
logger.setLevel('INFO')


def prepare_model_for_quantization(config, model, loader):
    quant, act = config.quant, config.act_quant
    pass_data_for_range_estimation(loader=loader, model=model, act_quant=quant.act_quant,
                                   weight_quant=quant.weight_quant, max_num_batches=act.num_batches,
                                   cross_entropy_layer=act.cross_entropy_layer)

    qat = config.qat
    if qat.learn_ranges:
        logger.info('Make quantizers learnable')
        model.learn_ranges()
    else:
        logger.info(f'Fix quantizer ranges to fixW={qat.fix_weight_ranges} and '
                    f'fixA={qat.fix_act_ranges}')
        model.estimate_ranges_train()          # ranges keep following the data while training ...
        for wanted, freeze in ((qat.fix_weight_ranges, model.fix_weight_ranges),
                               (qat.fix_act_ranges, model.fix_act_ranges)):
            if wanted:                         # ... unless asked to freeze them
                freeze()

    model.set_quant_state(quant.weight_quant, quant.act_quant)
    return model
