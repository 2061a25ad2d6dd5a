"""AdaRound (learned weight rounding) on the gfx950 kernels K10/K11/K13."""
from quantization.adaround.adaround import apply_adaround_to_layer
from quantization.adaround.utils import (
    AdaRoundInitMode,
    AdaRoundMode,
    AdaRoundActQuantMode,
    AdaRoundLossType,
    AdaRoundTempDecayType,
)
