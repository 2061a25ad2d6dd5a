"""Per-tensor quantization state machine: one quantizer + one range estimator.

Same surface as the reference's ``quantization/quantization_manager.py`` (``Qstates``,
``QuantizationManager`` with ``state / quantizer / range_estimator / axis / n_groups / n_bits``
and the ``estimate_ranges / fix_ranges / learn_ranges / estimate_ranges_train / reset_ranges /
set_quant_range / forward`` methods, reference :19-112).  In the estimating states a forward is

    estimator(x)  ->  set_quant_range  ->  quantizer(x)

i.e. ``tq_minmax`` (+ finalize) -> ``tq_range_update`` -> ``tq_set_range_*`` -> ``tq_fake_quant_fwd``
on the current HIP stream with no host synchronisation; in ``fix_ranges`` only the last launch runs.
"""
from enum import Enum

from torch import nn

from quantization.quantizers import QMethods, QuantizerNotInitializedError
from quantization.range_estimators import RangeEstimators


class Qstates(Enum):
    estimate_ranges = 0        # ranges follow the data in train and eval mode
    fix_ranges = 1             # ranges are frozen
    learn_ranges = 2           # quantizer parameters are nn.Parameters
    estimate_ranges_train = 3  # ranges follow the data in train mode only


class QuantizationManager(nn.Module):
    """Owns the quantizer selected by `qmethod` and the estimator selected by `init`.

    Parameters
    ----------
    qmethod : QMethods member
    init : RangeEstimators member
    per_channel : bool        one range per index of dim 0
    axis, n_groups            per-embedding / per-embedding-group activation quantization
    x_min, x_max              optional fixed range (skips estimation)
    qparams : dict            forwarded to the quantizer (n_bits, scale_domain, ...)
    init_params : dict        forwarded to the estimator (momentum, num_candidates, ...)
    """

    def __init__(self, qmethod=QMethods.symmetric_uniform, init=RangeEstimators.current_minmax,
                 per_channel=False, axis=None, n_groups=None, x_min=None, x_max=None, qparams=None,
                 init_params=None):
        super().__init__()
        self.state = Qstates.estimate_ranges
        self.qmethod = qmethod
        self.init = init
        self.per_channel = per_channel
        self.axis = axis
        self.n_groups = n_groups
        self.qparams = qparams if qparams else {}
        self.init_params = init_params if init_params else {}
        self.range_estimator = None

        self.quantizer = self.qmethod.cls(per_channel=per_channel, axis=axis, **self.qparams)

        if x_min is not None and x_max is not None:
            self.set_quant_range(x_min, x_max)
            self.state = Qstates.fix_ranges
        else:
            self.range_estimator = self.init.cls(
                per_channel=self.per_channel, quantizer=self.quantizer, axis=self.axis,
                n_groups=self.n_groups, **self.init_params)

    @property
    def n_bits(self):
        return self.quantizer.n_bits

    def estimate_ranges(self):
        self.state = Qstates.estimate_ranges

    def fix_ranges(self):
        if not self.quantizer.is_initialized:
            raise QuantizerNotInitializedError()
        self.state = Qstates.fix_ranges

    def learn_ranges(self):
        self.quantizer.make_range_trainable()
        self.state = Qstates.learn_ranges

    def estimate_ranges_train(self):
        self.state = Qstates.estimate_ranges_train

    def reset_ranges(self):
        self.range_estimator.reset()
        self.quantizer.reset()
        self.estimate_ranges()

    def _estimating(self):
        return self.state == Qstates.estimate_ranges or (
            self.state == Qstates.estimate_ranges_train and self.training)

    def forward(self, x):
        est = self.range_estimator
        if est is not None and est.per_group_range_estimation:
            est(x)          # PEG phase 1: only collect per-dimension ranges, pass x through
            return x
        if self._estimating():
            if est is None:
                raise RuntimeError('this manager was built with a fixed range: no estimator to run')
            cur_xmin, cur_xmax = est(x)
            self.set_quant_range(cur_xmin, cur_xmax)
        return self.quantizer(x)

    def set_quant_range(self, x_min, x_max):
        self.quantizer.set_quant_range(x_min, x_max)

    def extra_repr(self):
        return 'state={}'.format(self.state.name)
