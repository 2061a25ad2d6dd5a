"""Calibration driver and small helpers (counterpart of the hot-path part of the reference's
utils/utils.py: ``pass_data_for_range_estimation`` :47-79, ``seed_all`` :16-24, ``DotDict`` :82-103,
``Stopwatch`` :106-179, ``StopForwardException`` :42-44).

``pass_data_for_range_estimation`` is the loop that gets sharded over the GPUs of a node: with
``quantization.distributed`` enabled every rank feeds its slice of each calibration batch and the
range statistics are all-reduced inside the estimators.
"""
import os
import random
import time

import numpy as np
import torch
import torch.nn as nn

from quantization import distributed as tq_dist
from quantization.range_estimators import RangeEstimators


def seed_all(seed=1029):
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def count_params(module):
    return sum(p.numel() for p in module.parameters())


def count_embedding_params(model):
    return sum(count_params(m) for m in model.modules() if isinstance(m, nn.Embedding))


def get_layer_by_name(model, layer_name):
    return dict(model.named_modules()).get(layer_name, None)


class StopForwardException(Exception):
    """Thrown by hooks to abort a forward pass once the wanted tensor has been captured."""


def _install_cross_entropy_estimator(model, layer_name):
    layer = get_layer_by_name(model, layer_name)
    if not layer:
        raise ValueError('Cross-entropy layer not found')
    print(f'Set cross entropy estimator for layer "{layer_name}"')
    mgr = layer.activation_quantizer
    mgr.range_estimator = RangeEstimators.cross_entropy.cls(
        per_channel=mgr.per_channel, quantizer=mgr.quantizer, **mgr.init_params)


def pass_data_for_range_estimation(loader, model, act_quant, weight_quant, max_num_batches=20,
                                   cross_entropy_layer=None, inp_idx=0):
    """Run up to `max_num_batches` batches through `model` in eval mode so that every
    QuantizationManager in an estimating state sees data (one batch is enough when only the
    weights are quantized)."""
    model.set_quant_state(weight_quant, act_quant)
    model.eval()

    if cross_entropy_layer is not None:
        _install_cross_entropy_estimator(model, cross_entropy_layer)

    device = next(model.parameters()).device
    from quantization import options
    if weight_quant and options.LOCKSTEP_WEIGHT_SEARCH:
        # weight ranges do not depend on the data: the golden-section searches of all layers (README recipe) in lock step
        from quantization.autoquant_utils import precalibrate_weights
        with torch.no_grad():
            precalibrate_weights(model)
    for i, data in enumerate(loader):
        try:
            if isinstance(data, (tuple, list)):
                model(tq_dist.shard_batch(data[inp_idx]).to(device=device))
            else:
                model(**{k: tq_dist.shard_batch(v).to(device=device) for k, v in data.items()})
        except StopForwardException:
            pass

        if i >= max_num_batches - 1 or not act_quant:
            break
    tq_dist.check_exchange_health()      # sharded calibration over the P2P mailbox: a timed-out exchange is an error


class DotDict(dict):
    """dict with attribute access: ``cfg.a`` == ``cfg['a']``."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(f"DotDict instance has no key '{key}' ({self.keys()})") from None

    def __setattr__(self, key, value):
        self[key] = value

    def __delattr__(self, key):
        del self[key]


class Stopwatch:
    """Accumulating wall-clock timer usable as a context manager."""

    def __init__(self, name=None, verbose=False):
        self._name = name
        self._verbose = verbose
        self._t0 = 0.0
        self._total = 0.0
        self._running = False

    def __enter__(self):
        return self.start()

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.stop()
        if self._verbose:
            self.print()

    def start(self):
        if not self._running:
            self._t0 = time.time()
            self._running = True
        return self

    def stop(self):
        if self._running:
            self._total += time.time() - self._t0
            self._running = False
        return self

    def reset(self):
        self._t0 = self._total = 0.0
        self._running = False
        return self

    def get_total_duration(self):
        if self._running:
            now = time.time()
            self._total += now - self._t0
            self._t0 = now
        return self._total

    def format(self):
        prefix = f'[{self._name}]' if self._name is not None else 'Elapsed time'
        return f'{prefix}: {self.get_total_duration():.3f} sec'

    def print(self):
        print(self.format())
