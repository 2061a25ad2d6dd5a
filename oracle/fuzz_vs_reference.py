#!/usr/bin/env python3
"""Differential fuzz of the oracle against the REFERENCE ITSELF (build container only: needs /root/reference).

Test infrastructure, like everything under oracle/: it pins oracle/tq_oracle.py beyond the committed fixtures by
running the reference's own classes (quantization/quantizers.py, quantization/range_estimators.py) next to the
restatement on thousands of random configurations -- bit widths 2..16, symmetric / asymmetric, degenerate and
one-sided ranges, magnitudes 1e-6..1e6, NaN / +-inf / denormal / -0 data; per-tensor, per-channel, per-axis,
per-group and range-permuted group statistics through three-batch current / all-time / running min-max traces; MSE and
cross-entropy searches (1-D / 2-D grids, scipy golden section, per-channel rows) over two batches; per-channel and
per-axis parameter vectors; AdaRound alpha initialisation, soft / hard forward, regulariser and gradients in the three
rounding modes -- and demanding bit equality of parameters, indices, dequantised values and estimator state.

    python oracle/fuzz_vs_reference.py [n_quantizer_cases] [n_estimator_cases] [n_search_cases] [n_layout_cases] [n_adaround_cases]
                                                                      (defaults 4000, 1500, 300, 1000, 600)

Run in its own process: the reference's package is also called `quantization`.  tests/test_oracle_golden.py runs a
short version when /root/reference exists and skips otherwise (the GPU box has no reference).
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
_u = types.ModuleType('utils')
_u.__path__ = [os.path.join(REF, 'utils')]
sys.modules['utils'] = _u

import numpy as np  # noqa: E402
import torch  # noqa: E402
from quantization.quantizers import AsymmetricUniformQuantizer as RA, SymmetricUniformQuantizer as RS  # noqa: E402
from quantization.range_estimators import (  # noqa: E402
    AllMinMaxEstimator as RAll, CrossEntropyEstimator as RXent, CurrentMinMaxEstimator as RC, MSE_Estimator as RMSE,
    OptMethod as ROpt, RunningMinMaxEstimator as RRun)

from quantization.adaround.quantizer import (  # noqa: E402
    AdaRoundAsymmetricUniformQuantizer as RAdaA, AdaRoundSymmetricUniformQuantizer as RAdaS)
from quantization.adaround.utils import AdaRoundMode as RMode, CombinedLoss as RLoss  # noqa: E402

sys.path.insert(0, ROOT)
from oracle import tq_oracle as O  # noqa: E402

SPECIALS = [0.0, -0.0, float('inf'), -float('inf'), float('nan'), 1e-38, -1e-38, 1e38, -1e38, 3.4e38, 1e-45]


def _same(a, b):
    return torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(
        torch.nan_to_num(a, nan=7.0).view(torch.int32), torch.nan_to_num(b, nan=7.0).view(torch.int32))


def fuzz_quantizers(n_cases, seed=0):
    rs = np.random.RandomState(seed)
    bad = 0
    for _ in range(n_cases):
        n_bits = int(rs.choice([2, 3, 4, 6, 8, 16]))
        sym = bool(rs.randint(2))
        mag = 10.0 ** rs.uniform(-6, 6)
        shape = tuple(int(v) for v in rs.randint(1, 9, size=rs.randint(1, 4)))
        x = torch.tensor(rs.randn(*shape).astype(np.float32) * mag)
        if rs.rand() < 0.3:
            flat = x.reshape(-1)
            for _k in range(rs.randint(1, 4)):
                flat[rs.randint(flat.numel())] = SPECIALS[rs.randint(len(SPECIALS))]
        kind = rs.randint(5)
        lo, hi = sorted((rs.randn() * mag, rs.randn() * mag))
        if kind == 0:
            lo = 0.0
        elif kind == 1:
            hi = 0.0
        elif kind == 2:
            lo = hi = 0.0
        elif kind == 3:
            lo, hi = abs(lo), abs(lo) + abs(hi)
        Q = (RS if sym else RA)(n_bits=n_bits)
        Q.set_quant_range(torch.tensor(lo, dtype=torch.float32), torch.tensor(hi, dtype=torch.float32))
        yr, ir = Q(x), Q.to_integer_forward(x)
        if sym:
            d, signed = O.sym_params_from_range(torch.tensor(lo), torch.tensor(hi), n_bits)
            io, yo = O.fake_quant(x, d, None, n_bits, True, bool(signed))
            okp = torch.equal(d.reshape(()), Q._delta.reshape(())) and bool(signed) == bool(Q._signed)
        else:
            d, zf = O.asym_params_from_range(torch.tensor(lo), torch.tensor(hi), n_bits)
            io, yo = O.fake_quant(x, d, zf, n_bits, False)
            okp = torch.equal(d.reshape(()), Q._delta.reshape(())) and torch.equal(zf.reshape(()), Q._zero_float.reshape(()))
        if not (okp and _same(yr, yo) and _same(ir, io)):
            bad += 1
            if bad <= 8:
                print('MISMATCH', dict(n_bits=n_bits, sym=sym, lo=lo, hi=hi, shape=shape, params_ok=okp))
    return bad


def fuzz_estimators(n_cases, seed=1):
    rs = np.random.RandomState(seed)
    bad = 0
    for _ in range(n_cases):
        B, T, D = int(rs.randint(1, 5)), int(rs.randint(1, 9)), int(rs.choice([6, 12, 24]))
        mode = rs.randint(5)        # 0 per-tensor, 1 per-channel, 2 per-axis, 3 groups, 4 groups ordered by phase-1 ranges
        axis, n_groups, per_channel = None, None, mode == 1
        if mode >= 2:
            axis = 2
        if mode >= 3:
            n_groups = int(rs.choice([2, 3, 6]))
        xs = [torch.tensor((rs.randn(B, T, D) * 10 ** rs.uniform(-2, 2)).astype(np.float32)) for _k in range(3)]
        kind = rs.randint(3)
        q = RA(n_bits=8, axis=axis, per_channel=per_channel)
        kw = dict(per_channel=per_channel, quantizer=q, axis=axis, n_groups=n_groups)
        mom = float(rs.choice([0.9, 0.5, 0.99]))
        est = RC(**kw) if kind == 0 else (RAll(**kw) if kind == 1 else RRun(momentum=mom, **kw))
        ranges = None
        if mode == 4 and kind == 0:
            est.per_group_range_estimation = True
            est(xs[0])
            est.per_group_range_estimation = False
            ranges = est.ranges.clone()
        cur = None
        for x in xs:
            rmin, rmax = est(x)
            if kind == 1:           # AllMinMax ignores axis / groups
                bm, bM = O.batch_minmax(x, axis=None, n_groups=None, per_channel=per_channel)
                cur = (bm, bM) if cur is None else O.allminmax_update(cur[0], cur[1], bm, bM)
            elif kind == 0:
                cur = O.batch_minmax(x, axis=axis, n_groups=n_groups, per_channel=per_channel, ranges=ranges)
            else:
                bm, bM = O.batch_minmax(x, axis=axis, n_groups=n_groups, per_channel=per_channel)
                cur = (bm, bM) if cur is None else O.running_update(cur[0], cur[1], bm, bM, mom)
            if not (torch.equal(rmin.reshape(-1), cur[0].reshape(-1)) and torch.equal(rmax.reshape(-1), cur[1].reshape(-1))):
                bad += 1
                if bad <= 8:
                    print('MISMATCH', dict(kind=kind, mode=mode, n_groups=n_groups, shape=(B, T, D)))
                break
    return bad


def fuzz_layouts(n_cases, seed=3):
    """Per-channel (dim 0) and per-axis parameter vectors: set_quant_range with vectors, forward, to_integer_forward."""
    rs = np.random.RandomState(seed)
    bad = 0
    for _ in range(n_cases):
        n_bits = int(rs.choice([2, 4, 8]))
        per_axis = bool(rs.randint(2))
        if per_axis:            # [B, T, d] activations, one range per embedding dimension (asymmetric only, quirk q3)
            shape, axis, sym, n = (int(rs.randint(1, 4)), int(rs.randint(1, 6)), int(rs.choice([4, 12]))), 2, False, None
            n = shape[2]
            Q = RA(n_bits=n_bits, axis=2)
        else:                   # [C, K] weights, one range per output channel
            shape, axis, sym = (int(rs.randint(2, 9)), int(rs.choice([3, 16]))), None, bool(rs.randint(2))
            n = shape[0]
            Q = (RS if sym else RA)(n_bits=n_bits, per_channel=True)
        x = torch.tensor((rs.randn(*shape) * 10 ** rs.uniform(-2, 2)).astype(np.float32))
        lo = torch.tensor((-np.abs(rs.randn(n)) * 2).astype(np.float32))
        hi = torch.tensor((np.abs(rs.randn(n)) * 2).astype(np.float32))
        if rs.randint(3) == 0:
            lo[rs.randint(n)] = 0.0
        Q.set_quant_range(lo, hi)
        yr, ir = Q(x), Q.to_integer_forward(x)
        if sym:
            d, signed = O.sym_params_from_range(lo, hi, n_bits)
            io, yo = O.fake_quant(x, d, None, n_bits, True, bool(signed), per_channel=True)
        else:
            d, zf = O.asym_params_from_range(lo, hi, n_bits)
            io, yo = O.fake_quant(x, d, zf, n_bits, False, axis=axis, per_channel=not per_axis)
        if not (_same(yr, yo) and _same(ir, io)):
            bad += 1
            if bad <= 8:
                print('MISMATCH', dict(n_bits=n_bits, per_axis=per_axis, sym=sym, shape=shape))
    return bad


def fuzz_adaround(n_cases, seed=4):
    """AdaRoundQuantizer: alpha initialisation, soft and hard forward, the regulariser's value and its gradient
    together with the gradient of the soft forward w.r.t. alpha (autograd on both sides), three rounding modes."""
    rs = np.random.RandomState(seed)
    modes = {'learned_sigmoid': RMode.learned_sigmoid, 'learned_hard_sigmoid': RMode.learned_hard_sigmoid,
             'sigmoid_temp_decay': RMode.sigmoid_temp_decay}
    bad = 0
    for _ in range(n_cases):
        n_bits = int(rs.choice([2, 4, 8]))
        sym = bool(rs.randint(2))
        name = list(modes)[rs.randint(3)]
        temp = float(rs.choice([0.5, 1.0, 2.0])) if name == 'sigmoid_temp_decay' else None
        w = torch.tensor((rs.randn(int(rs.randint(2, 9)), int(rs.choice([3, 16]))) * 10 ** rs.uniform(-1, 1)).astype(np.float32))
        lo, hi = float(w.min()) * rs.uniform(0.5, 1.1), float(w.max()) * rs.uniform(0.5, 1.1)
        Q = (RAdaS if sym else RAdaA)(n_bits=n_bits)
        Q.set_quant_range(torch.tensor(lo, dtype=torch.float32), torch.tensor(hi, dtype=torch.float32))
        Q.round_mode, Q.temperature = modes[name], temp
        Q.soft_targets = True
        ys = Q(w)                                      # first call initialises alpha
        alpha_r = Q.alpha.detach().clone()
        Q.soft_targets = False
        yh = Q(w)
        if sym:
            d, signed = O.sym_params_from_range(torch.tensor(lo), torch.tensor(hi), n_bits)
            zf, signed = None, bool(signed)
        else:
            (d, zf), signed = O.asym_params_from_range(torch.tensor(lo), torch.tensor(hi), n_bits), False
        alpha_o = O.ada_alpha_init(w, O.effective_scale(d), name, temp)
        ok = _same(alpha_r, alpha_o)
        a = alpha_r.clone().requires_grad_(True)       # same alpha on both sides from here on
        _, ys_o = O.ada_fake_quant(w, a, d, zf, n_bits, sym, signed, name, True, temperature=temp)
        _, yh_o = O.ada_fake_quant(w, a, d, zf, n_bits, sym, signed, name, False, temperature=temp)
        ok = ok and _same(ys.detach(), ys_o.detach()) and _same(yh.detach(), yh_o.detach())
        b, weight = float(rs.uniform(2, 20)), 0.01
        g = torch.tensor(rs.randn(*w.shape).astype(np.float32))
        Q.soft_targets = True
        Q.alpha.grad = None
        reg_r = weight * (1 - ((Q.get_rest().view(-1) - 0.5).abs() * 2).pow(b)).sum()       # adaround/utils.py:159-162
        ((Q(w) * g).sum() + reg_r).backward()
        reg_o = O.ada_round_reg(a, name, b, weight, temp)
        ((ys_o * g).sum() + reg_o).backward()
        ok = ok and _same(reg_r.detach(), reg_o.detach()) and _same(Q.alpha.grad, a.grad)
        if not ok:
            bad += 1
            if bad <= 8:
                print('MISMATCH', dict(n_bits=n_bits, sym=sym, mode=name, temp=temp, shape=tuple(w.shape)))
    return bad


def fuzz_searches(n_cases, seed=2):
    """MSE_Estimator / CrossEntropyEstimator (1-D and 2-D grids, golden section, per-channel rows) over two batches:
    returned thresholds and the accumulated fp64 loss array, bit for bit."""
    rs = np.random.RandomState(seed)
    bad = 0
    for _ in range(n_cases):
        n_bits = int(rs.choice([2, 3, 4, 8]))
        sym = bool(rs.randint(2))
        per_channel = bool(rs.randint(3) == 0)
        golden = bool(rs.randint(3) == 0)
        xent = (not per_channel) and rs.randint(5) == 0
        C = int(rs.choice([5, 10, 20]))
        rows, cols = int(rs.randint(2, 7)), int(rs.choice([2, 16, 40]))
        mag = 10.0 ** rs.uniform(-2, 2)
        datas = [torch.tensor((rs.randn(rows, cols) * mag).astype(np.float32)) for _k in range(2)]
        if rs.randint(4) == 0:
            datas = [d.abs() for d in datas]                 # one-sided
        q = (RS if sym else RA)(n_bits=n_bits, per_channel=per_channel)
        opt = ROpt.golden_section if golden else ROpt.grid
        est = (RXent if xent else RMSE)(per_channel=per_channel, quantizer=q, num_candidates=C, opt_method=opt)
        o = O.MSESearch(O.QSpec(n_bits, sym), num_candidates=C, opt_method='golden_section' if golden else 'grid',
                        per_channel=per_channel, loss_value=O.xent_loss_value if xent else O.mse_loss_value)
        ok = True
        for d in datas:
            rmin, rmax = est(d)
            omin, omax = o.step_batch(d)
            ok = ok and torch.equal(torch.as_tensor(rmin).reshape(-1).float(), omin.reshape(-1).float()) \
                and torch.equal(torch.as_tensor(rmax).reshape(-1).float(), omax.reshape(-1).float())
            if not golden:
                ok = ok and np.array_equal(np.asarray(est.loss_array), o.loss_array)
        if not ok:
            bad += 1
            if bad <= 8:
                print('MISMATCH', dict(n_bits=n_bits, sym=sym, per_channel=per_channel, golden=golden, xent=xent, C=C,
                                       shape=(rows, cols)))
    return bad


if __name__ == '__main__':
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    ne = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    ns = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    nl = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
    na = int(sys.argv[5]) if len(sys.argv) > 5 else 600
    bq, be_, bs, bl, ba = fuzz_quantizers(nq), fuzz_estimators(ne), fuzz_searches(ns), fuzz_layouts(nl), fuzz_adaround(na)
    print(f'quantizer cases {nq}: {bq} mismatches; estimator cases {ne}: {be_} mismatches; search cases {ns}: {bs} mismatches; '
          f'layout cases {nl}: {bl} mismatches; adaround cases {na}: {ba} mismatches')
    sys.exit(1 if (bq or be_ or bs or bl or ba) else 0)
