"""How far the opt-in INTEGER evaluation (options.INT8_LINEAR + the harness models' `fuse` switches: exact integer
contractions on the i8 matrix cores) moves a model away from the reference's contract, the fp32 SIMULATION of the same
quantized network (reference quantization/hijacker.py:66-70: `F.linear` on dequantised fp32 tensors) -- layer by layer.

Both evaluate the same quantized network; they differ by the simulation's fp32 accumulation round-off inside every GEMM
(the integer path computes the exact value the simulation approximates).  On coarse grids a difference of 1e-7 in a
pre-activation flips a rounding decision by a whole step, and a flipped index is an O(1) perturbation for everything
downstream.  Two numbers per encoder layer make that visible:

* `same_input`: the layer is fed the LAYERED trajectory's input in both modes -- the error one layer adds by itself;
* `free_running`: the integer forward runs on its own outputs -- what accumulates.

A "flip" is an output element whose index on the layer's output grid differs; `max_steps` is the largest index distance.
"""
import torch

from quantization.quantization_manager import QuantizationManager


def _last_activation_grid(layer, h, mask):
    """delta of the LAST activation quantizer `layer` calls (the grid its output lives on)."""
    called = []
    hooks = [m.register_forward_hook(lambda mod, i, o: called.append(mod))
             for n, m in layer.named_modules() if isinstance(m, QuantizationManager) and n.endswith('activation_quantizer')]
    try:
        layer(h, mask)
    finally:
        for hk in hooks:
            hk.remove()
    return float(called[-1].quantizer._delta.reshape(-1)[0])


def encoder_flip_rates(model, ids, integer_mode):
    """`model`: a calibrated, fixed-range harness model (`embeddings`, `layers`); `integer_mode`: a context manager that
    switches the integer evaluation on.  -> list of per-layer dicts + the index of the first layer whose free-running
    output differs from the layered one (None if none does)."""
    with torch.no_grad():
        mask = torch.zeros(ids.shape[0], 1, 1, ids.shape[1], device=ids.device)
        hs = [model.embeddings(ids)]
        grids = []
        for layer in model.layers:
            grids.append(_last_activation_grid(layer, hs[-1], mask))
            hs.append(layer(hs[-1], mask))
        with integer_mode:
            same = [layer(hs[i], mask) for i, layer in enumerate(model.layers)]
            free, h = [], hs[0]
            for layer in model.layers:
                h = layer(h, mask)
                free.append(h)
    rows, first = [], None
    for i, delta in enumerate(grids):
        ref = torch.round(hs[i + 1].double() / delta)
        row = {'layer': i, 'output_grid_step': delta}
        for key, y in (('same_input', same[i]), ('free_running', free[i])):
            d = (torch.round(y.double() / delta) - ref).abs()
            row[key] = {'flip_rate': float((d > 0).double().mean()), 'max_steps': int(d.max())}
        if first is None and row['free_running']['flip_rate'] > 0:
            first = i
        rows.append(row)
    return rows, first
