"""Hot-path helpers of the reference's ``utils`` package (calibration loop, per-embedding wiring,
AdaRound / QAT drivers).  The GLUE / HuggingFace / click harness of the reference is out of scope
(SURVEY.md section 2 rows 16-21).

``quantization.adaround`` imports ``utils.utils`` and ``utils.adaround_utils`` imports
``quantization.adaround``; the AdaRound / QAT drivers are therefore resolved lazily so that either
package can be imported first."""
from utils.per_embd_quant_utils import (
    hijack_act_quant,
    hijack_weight_quant,
    hijack_act_quant_modules,
    set_act_quant_axis_and_groups,
)
from utils.utils import (
    seed_all,
    count_params,
    count_embedding_params,
    get_layer_by_name,
    pass_data_for_range_estimation,
    DotDict,
    Stopwatch,
    StopForwardException,
)

_LAZY = {
    'apply_adaround_to_model': 'utils.adaround_utils',
    'get_train_samples': 'utils.adaround_utils',
    'prepare_model_for_quantization': 'utils.qat_utils',
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module(_LAZY[name]), name)
    raise AttributeError(f"module 'utils' has no attribute '{name}'")
