"""Default AdaRound hyper-parameters (same keys/values as the reference's
quantization/adaround/config.py:17-38)."""
from quantization.adaround.utils import (
    AdaRoundActQuantMode,
    AdaRoundInitMode,
    AdaRoundMode,
    AdaRoundTempDecayType,
)
from utils.utils import DotDict


class AdaRoundConfig(DotDict):
    pass


DEFAULT_ADAROUND_CONFIG = AdaRoundConfig(
    layers=('all',),                                  # which layers to optimise
    num_samples=1024,                                 # cached calibration samples
    init=AdaRoundInitMode.range_estimator,            # weight grid initialisation
    round_mode=AdaRoundMode.learned_hard_sigmoid,     # relaxation h(alpha)
    asym=True,                                        # layer input from the quantized network
    include_act_func=True,
    lr=1e-3,
    iters=1000,
    weight=0.01,                                      # lambda of the rounding regulariser
    annealing=(20, 2),                                # beta: start -> end
    decay_type=AdaRoundTempDecayType.cosine,
    decay_shape=1.0,
    decay_start=0.0,
    warmup=0.2,                                       # fraction of iters without regulariser
    act_quant_mode=AdaRoundActQuantMode.post_adaround,
)
