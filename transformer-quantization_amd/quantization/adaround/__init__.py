"""AdaRound (learned weight rounding) on the gfx950 kernels K10/K11/K13."""
from quantization.adaround import utils as _options
from quantization.adaround.adaround import apply_adaround_to_layer

AdaRoundInitMode = _options.AdaRoundInitMode
AdaRoundMode = _options.AdaRoundMode
AdaRoundActQuantMode = _options.AdaRoundActQuantMode
AdaRoundLossType = _options.AdaRoundLossType
AdaRoundTempDecayType = _options.AdaRoundTempDecayType

__all__ = ['apply_adaround_to_layer', 'AdaRoundInitMode', 'AdaRoundMode', 'AdaRoundActQuantMode',
           'AdaRoundLossType', 'AdaRoundTempDecayType']
