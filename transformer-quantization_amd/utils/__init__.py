"""Hot-path helpers of the reference's ``utils`` package (calibration loop, per-embedding wiring,
AdaRound / QAT drivers).  The GLUE / HuggingFace / click harness of the reference is out of scope
(SURVEY.md section 2 rows 16-21)."""
from utils.adaround_utils import apply_adaround_to_model, get_train_samples
from utils.per_embd_quant_utils import (
    hijack_act_quant,
    hijack_weight_quant,
    hijack_act_quant_modules,
    set_act_quant_axis_and_groups,
)
from utils.qat_utils import prepare_model_for_quantization
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
