"""Pin the oracle (oracle/tq_oracle.py) against the golden vectors captured from the reference.

CPU only.  Bit-exact: the oracle performs the same ATen ops in the same order as the reference.
"""
import numpy as np
import pytest
import torch

from oracle import tq_oracle as O
from tests._cases import fq_case, est_inputs, t

torch.set_num_threads(1)


def _params_from_range(c):
    """oracle range -> (delta, zero_float, signed) for a fake_quant case."""
    vec = c['axis'] is not None or c['per_channel']
    if c['symmetric']:
        delta, signed = O.sym_params_from_range(c['xmin'], c['xmax'], c['n_bits'],
                                                allow_vector=vec)
        return delta, None, bool(signed.item())
    delta, zf = O.asym_params_from_range(c['xmin'], c['xmax'], c['n_bits'], allow_vector=vec)
    return delta, zf, False


def test_fake_quant_cases_bit_exact(golden_fake_quant):
    z, meta = golden_fake_quant
    assert len(meta) >= 60
    for m in meta:
        c = fq_case(z, m)
        x = c['x'].float()
        # range statistics
        mn, mx = O.batch_minmax(x, c['axis'], c['n_groups'], c['per_channel'], c['ranges'])
        assert torch.equal(mn.reshape(-1), c['xmin'].reshape(-1)), m
        assert torch.equal(mx.reshape(-1), c['xmax'].reshape(-1)), m
        # range -> params
        delta, zf, signed = _params_from_range(c)
        assert torch.equal(delta.reshape(-1), c['delta'].reshape(-1)), m
        if zf is not None:
            assert torch.equal(zf.reshape(-1), c['zero_float'].reshape(-1)), m
        if c['symmetric']:
            assert signed == m['signed']
        lo, hi = O.grid_limits(c['n_bits'], c['symmetric'], signed)
        assert (lo, hi) == (m['int_min'], m['int_max'])
        # the op
        idx, y = O.fake_quant(x, delta, zf, c['n_bits'], c['symmetric'], signed,
                              axis=c['axis'], per_channel=c['per_channel'])
        assert torch.equal(idx, c['idx']), m
        assert torch.equal(y, c['y']), m
        if m['io'] == 'bf16':
            _, yb = O.fake_quant_lowp(c['x'], delta, zf, c['n_bits'], c['symmetric'], signed,
                                      axis=c['axis'], per_channel=c['per_channel'])
            assert torch.equal(yb, c['y_bf16']), m


def _drive_estimator(m, xs):
    """Replay an estimator trace with oracle functions; returns lists per batch."""
    from tests._cases import LAYOUT_ARGS
    la = LAYOUT_ARGS[m['layout']]
    sym = m['method'] == 'symmetric_uniform'
    init, ip = m['init'], m['init_params']
    vec = la['axis'] is not None or la['per_channel']
    mins, maxs, deltas, zfs = [], [], [], []
    cur = (None, None)
    search = None
    y = None
    for x in xs:
        if init in ('current_minmax', 'allminmax', 'running_minmax'):
            if init == 'allminmax':   # ignores axis / groups (quirk q5)
                new = O.batch_minmax(x, None, None, la['per_channel'])
                cur = O.allminmax_update(cur[0], cur[1], *new)
            else:
                new = O.batch_minmax(x, la['axis'], la['n_groups'], la['per_channel'])
                cur = new if init == 'current_minmax' else O.running_update(
                    cur[0], cur[1], *new, momentum=ip.get('momentum', 0.9))
        else:
            if search is None:
                q = O.QSpec(m['n_bits'], sym, axis=la['axis'])
                search = O.MSESearch(
                    q, num_candidates=ip.get('num_candidates', 100),
                    opt_method=ip.get('opt_method', 'grid'), per_channel=la['per_channel'],
                    loss_value=O.xent_loss_value if init == 'cross_entropy' else O.mse_loss_value)
            cur = search.step_batch(x)
            # the reference shares one quantizer between manager and estimator
            search.q.set_range(cur[0], cur[1])
        if sym:
            delta, signed = O.sym_params_from_range(cur[0], cur[1], m['n_bits'], allow_vector=True)
            zf, signed = None, bool(signed.item())
        else:
            delta, zf = O.asym_params_from_range(cur[0], cur[1], m['n_bits'], allow_vector=True)
            signed = False
        _, y = O.fake_quant(x, delta, zf, m['n_bits'], sym, signed, axis=la['axis'],
                            per_channel=la['per_channel'])
        mins.append(cur[0].reshape(-1).clone())
        maxs.append(cur[1].reshape(-1).clone())
        deltas.append(delta.reshape(-1).clone())
        if zf is not None:
            zfs.append(zf.reshape(-1).clone())
    return mins, maxs, deltas, zfs, y, search


def test_estimator_traces_bit_exact(golden_estimators):
    z, meta = golden_estimators
    assert len(meta) >= 30
    for m in meta:
        k = m['k']
        xs = est_inputs(z, m)
        mins, maxs, deltas, zfs, y, search = _drive_estimator(m, xs)
        assert torch.equal(torch.stack(mins), t(z[f'e{k}_xmin'])), m
        assert torch.equal(torch.stack(maxs), t(z[f'e{k}_xmax'])), m
        if f'e{k}_delta' in z.files:
            assert torch.equal(torch.stack(deltas), t(z[f'e{k}_delta'])), m
        if zfs and f'e{k}_zero_float' in z.files:
            assert torch.equal(torch.stack(zfs), t(z[f'e{k}_zero_float'])), m
        assert torch.equal(y, t(z[f'e{k}_y_last'])), m
        if f'e{k}_loss_array' in z.files and search is not None and search.loss_array is not None:
            assert np.array_equal(search.loss_array, z[f'e{k}_loss_array']), m


def test_permuted_peg(golden_estimators):
    z, _ = golden_estimators
    xs = [t(b) for b in z['batches']]
    ranges = None
    for i, x in enumerate(xs):
        ranges = O.axis_ranges(x, 2, first=(i == 0))       # last batch wins (quirk q4)
    assert torch.equal(ranges, t(z['perm_ranges']))
    mn, mx = O.minmax_groups(xs[0], 2, 4, ranges)
    assert torch.equal(mn, t(z['perm_xmin']))
    assert torch.equal(mx, t(z['perm_xmax']))
    delta, zf = O.asym_params_from_range(mn, mx, 8)
    _, y = O.fake_quant(xs[0], delta, zf, 8, False, axis=2)
    assert torch.equal(y, t(z['perm_y']))


def test_adaround_quantizer_and_trace(golden_adaround):
    z, meta = golden_adaround
    for m in meta:
        k = m['k']
        sym = m['method'] == 'symmetric_uniform'
        w, b = t(z[f'a{k}_w']), t(z[f'a{k}_b'])
        X, tgt = t(z[f'a{k}_X']), t(z[f'a{k}_tgt'])
        delta = t(z[f'a{k}_delta'])
        zf = None if sym else t(z[f'a{k}_zero_float'])
        mode = m['mode']
        temp = 20
        scale = O.effective_scale(delta)
        alpha0 = O.ada_alpha_init(w, scale, mode, temp)
        assert torch.equal(alpha0, t(z[f'a{k}_alpha0'])), m
        args = (delta, zf, m['n_bits'], sym, bool(m['signed']), mode)
        _, soft0 = O.ada_fake_quant(w, alpha0, *args, soft=True, temperature=temp)
        idx0, hard0 = O.ada_fake_quant(w, alpha0, *args, soft=False, temperature=temp)
        assert torch.equal(soft0, t(z[f'a{k}_wq_soft0'])), m
        assert torch.equal(hard0, t(z[f'a{k}_wq_hard0'])), m
        assert torch.equal(idx0, t(z[f'a{k}_idx_hard0'])), m
        # optimisation trace with the recorded batch indices
        alpha = alpha0.clone().requires_grad_(True)
        opt = torch.optim.Adam([alpha], lr=m['lr'])
        batch_idx = z[f'a{k}_batch_idx']
        for it in range(m['iters']):
            idx = torch.from_numpy(batch_idx[it])
            opt.zero_grad()
            _, wq = O.ada_fake_quant(w, alpha, *args, soft=True, temperature=temp)
            out = torch.nn.functional.linear(X[idx], wq, b)
            if mode == 'sigmoid_temp_decay':
                # temp_decay loss type: no regulariser, temperature follows b (utils.py:154-157)
                rec = O.ada_rec_loss(out, tgt[idx])
                bval = O.temp_decay(it + 1, m['iters'], (20, 2), 0.2, 'cosine')
                loss = rec
                if (it + 1) >= m['iters'] * 0.2:
                    temp = bval
            else:
                loss, _ = O.ada_combined_loss(out, tgt[idx], alpha, it + 1, mode, 0.01,
                                              m['iters'], (20, 2), warmup=0.2)
            loss.backward()
            assert torch.equal(alpha.grad, t(z[f'a{k}_grads'][it])), (m, it)
            opt.step()
            assert float(loss) == z[f'a{k}_losses'][it], (m, it)
            assert torch.equal(alpha.detach(), t(z[f'a{k}_alphas'][it])), (m, it)
        _, hard1 = O.ada_fake_quant(w, alpha.detach(), *args, soft=False, temperature=temp)
        assert torch.equal(hard1, t(z[f'a{k}_wq_hard1'])), m


def test_ste_grad_matches_analytic():
    """dx of the op is the clamp mask."""
    torch.manual_seed(0)
    x = torch.randn(64, 32) * 3
    delta, zf = O.asym_params_from_range(torch.tensor(-2.0), torch.tensor(2.5), 4)
    y, dx, dd, dz = O.fake_quant_with_grads(x, delta, zf, 4, False)
    idx, _ = O.fake_quant(x, delta, zf, 4, False)
    zp = O.effective_zero_point(zf, 4)
    raw = torch.round(x / delta) + zp
    mask = ((raw >= 0) & (raw <= 15)).float()
    assert torch.allclose(dx, mask, atol=1e-6)
    assert dd.shape == delta.shape and dz.shape == zf.shape


def test_aten_sum_restatement():
    """oracle/aten_sum.py (numpy restatement of ATen's cascade sum, the order the reference's loss_fx values
    come out in) == the live torch.sum of this container, bit for bit, on the shapes the MSE estimator
    produces: row sums of weights / activations followed by the sum of the row sums."""
    from oracle import aten_sum as A
    g = torch.Generator().manual_seed(11)
    for n in (1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 31, 32, 33, 63, 100, 255, 257, 768, 769, 1000, 3072, 4097,
              8192 + 17, 30522, 32767):
        for _ in range(3):
            x = (torch.randn(n, generator=g) * 3) ** 2
            assert float(A.row_sum(x.numpy())) == torch.sum(x).item(), n
    for shape in ((768, 768), (8, 98304), (64, 3072), (768, 1), (2, 768), (5, 7), (100, 5), (3, 131072 + 555)):
        e = (torch.randn(*shape, generator=g) * 2) ** 2
        ref_rows = torch.sum(e, dim=1)
        assert np.array_equal(A.sum_rows(e.numpy()), ref_rows.numpy()), shape
        assert float(A.loss_sum(e.numpy())) == torch.sum(ref_rows).item(), shape
        assert np.array_equal(A.loss_sum(e.numpy(), per_channel_loss=True), ref_rows.numpy()), shape


def test_oracle_fuzz_against_the_reference_itself():
    """Build container only: oracle/fuzz_vs_reference.py runs the reference's own quantizer / estimator classes next to
    the restatement on random configurations (bit equality of parameters, indices, values, estimator state).  Its own
    process, because the reference's package is also called `quantization`; skipped where /root/reference is absent."""
    import os
    import subprocess
    import sys
    from tests.conftest import ROOT
    if not os.path.isdir('/root/reference/quantization'):
        pytest.skip('no /root/reference here (GPU box): the committed fixtures pin the oracle')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'fuzz_vs_reference.py'), '600', '250', '60', '300', '200'],
                       capture_output=True, text=True, cwd='/tmp', timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count(' 0 mismatches') == 5, r.stdout[-500:]
