"""Evaluation routes of the harness models and how far each one is from the REFERENCE.

A fixed-range forward can run
  * 'layered'     -- the reference's module chain: one launch per quantizer around torch's fp32 GEMMs (hipBLASLt);
                     what calibration and training always run;
  * 'fused_tails' -- the same GEMMs with the residual / LayerNorm (NoNorm) / softmax chains as single kernels;
  * 'integer'     -- exact integer GEMMs on the i8 matrix cores with fused epilogues, integer attention core, fused
                     tails (quantization/fused.py, options.INT8_LINEAR = True + every `fuse` switch);
  * 'default'     -- whatever the product does on its own (options.INT8_LINEAR = 'auto', `fuse = None`).

`compare_routes` judges them against outputs of the reference itself (tests/golden/*_hidden.npz: grid indices of the
encoder output after selected layers -- 0.5-0.8 M samples each -- and logits on four evaluation batches), not against
each other.  Used by tests/test_bert_e2e.py, tests/test_mobilebert_e2e.py, scripts/int_vs_reference.py.
"""
import numpy as np
import torch

from quantization import options

ROUTES = ('layered', 'fused_tails', 'integer', 'default')


class Route:
    """with Route(model, 'integer'): ...   -- switches are process-wide (class attributes + options), restored on exit."""

    def __init__(self, model, name):
        assert name in ROUTES, name
        self.name = name
        from harness import bert, mobilebert
        if isinstance(model, mobilebert.QMobileBertForSequenceClassification):
            self.tails = [(mobilebert.QResidualNoNorm, 'fuse')]
            self.integer = [(mobilebert.QBottleneckLayer, 'fuse'), (mobilebert.QFFN, 'fuse'),
                            (mobilebert.QMobileSelfAttention, 'fuse'), (mobilebert.QMobileLayer, 'fuse_ffn')]
        else:
            self.tails = [(bert.QResidualBlock, 'fuse'), (bert.QSelfAttention, 'fuse'), (bert.QEmbeddings, 'fuse')]
            self.integer = [(bert.QLayer, 'fuse_ffn')]

    def __enter__(self):
        self.saved = [(c, a, c.__dict__.get(a)) for c, a in self.tails + self.integer] + [options.INT8_LINEAR]
        n = self.name
        if n == 'default':
            for c, a in self.tails + self.integer:
                setattr(c, a, None)
            options.INT8_LINEAR = 'auto'
        else:
            for c, a in self.tails:
                setattr(c, a, n in ('fused_tails', 'integer'))
            for c, a in self.integer:
                setattr(c, a, n == 'integer')
            options.INT8_LINEAR = n == 'integer'
        return self

    def __exit__(self, *exc):
        options.INT8_LINEAR = self.saved.pop()
        for c, a, v in self.saved:
            setattr(c, a, v)
        return False


def install_reference_ranges(managers, ranges):
    """Give every activation quantizer the (xmin, xmax) the reference found at the same site (managers and ranges in the
    same order), so that the only difference left to the reference's forward is the arithmetic of the route."""
    for m, (lo, hi) in zip(managers, ranges):
        dev = m.quantizer._delta.device
        lo_t = torch.tensor(float(lo), dtype=torch.float32, device=dev)
        hi_t = torch.tensor(float(hi), dtype=torch.float32, device=dev)
        m.quantizer.set_quant_range(lo_t, hi_t)
        m.range_estimator.current_xmin, m.range_estimator.current_xmax = lo_t, hi_t
    options.invalidate_derived_caches()


def hidden_deviation(model, ids, zh, layers):
    """Encoder output after the given layers against the reference's, in steps of the REFERENCE's grid at that site:
    {L: {'same_grid_point_frac', 'mean_abs_dev_steps', 'max_abs_dev_steps'}} and the logits of the same forward."""
    got = {}
    hooks = [model.layers[k - 1].register_forward_hook(lambda m, i, o, k=k: got.__setitem__(k, o.detach())) for k in layers]
    try:
        logits = model(ids)
    finally:
        for h in hooks:
            h.remove()
    out = {}
    for k in layers:
        d = float(zh[f'hidden_delta_L{k}'])
        top = 2.0 ** int(np.ceil(np.log2(float(zh[f'hidden_idx_L{k}'].max()) + 1))) - 1          # 15 or 255
        zp = float(np.clip(np.rint(zh[f'hidden_zero_float_L{k}']), 0, top))
        ref = (zh[f'hidden_idx_L{k}'].astype(np.float64) - zp) * d
        dev = np.abs(got[k].double().cpu().numpy() - ref) / d
        out[f'L{k}'] = {'same_grid_point_frac': float((dev < 0.5).mean()), 'mean_abs_dev_steps': float(dev.mean()),
                        'max_abs_dev_steps': float(dev.max())}
    return out, logits


def logit_deviation(out, ref):
    d = np.abs(out.detach().double().cpu().numpy() - ref.astype(np.float64))
    span = float(ref.max() - ref.min())
    return {'max_abs': float(d.max()), 'mean_abs': float(d.mean()), 'max_over_span': float(d.max() / span),
            'mean_over_span': float(d.mean() / span),
            'argmax_agree': float((out.detach().cpu().numpy().argmax(-1) == ref.argmax(-1)).mean())}


def compare_routes(model, ids, zh, layers, routes=('layered', 'integer')):
    """{route: {'hidden': {...}, 'logits_4_batches': {...}, 'logits': tensor [32, n_labels]}} for a model whose ranges are
    fixed.  (The module-level forward hooks used to read the layer outputs sit on the encoder LAYER containers, which
    no fused launch skips.)"""
    res = {}
    ids = ids.to(next(model.parameters()).device)
    extra = torch.from_numpy(zh['input_ids_extra']).to(ids.device)
    ref_logits = np.concatenate([zh['logits']] + list(zh['logits_extra']))
    with torch.no_grad():
        for r in routes:
            with Route(model, r):
                hid, lo = hidden_deviation(model, ids, zh, layers)
                logits = torch.cat([lo] + [model(extra[i]) for i in range(extra.shape[0])])
            res[r] = {'hidden': hid, 'logits_4_batches': logit_deviation(logits, ref_logits), 'logits': logits}
    return res
