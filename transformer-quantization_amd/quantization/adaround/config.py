"""Default AdaRound hyper-parameters (same keys/values as the reference's
quantization/adaround/config.py:17-38)."""
from quantization.adaround import utils as opt
from utils.utils import DotDict


class AdaRoundConfig(DotDict):
    pass


_WHAT = dict(layers=('all',), num_samples=1024, init=opt.AdaRoundInitMode.range_estimator)
_RELAXATION = dict(round_mode=opt.AdaRoundMode.learned_hard_sigmoid, asym=True, include_act_func=True)
_OPTIMISER = dict(lr=1e-3, iters=1000)
_REGULARISER = dict(weight=0.01, annealing=(20, 2), warmup=0.2, decay_start=0.0, decay_shape=1.0,
                    decay_type=opt.AdaRoundTempDecayType.cosine)
_ACTIVATIONS = dict(act_quant_mode=opt.AdaRoundActQuantMode.post_adaround)

DEFAULT_ADAROUND_CONFIG = AdaRoundConfig(**_WHAT, **_RELAXATION, **_OPTIMISER, **_REGULARISER,
                                         **_ACTIVATIONS)
