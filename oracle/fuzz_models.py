#!/usr/bin/env python3
"""Module-level differential fuzz: the SAME script drives either the reference's `quantization` package or this
repository's drop-in (on the CPU, through the oracle-backed backend double of tests/_oracle_backend.py) -- which is the
drop-in claim itself -- on random small networks and quantization settings, and dumps every observable: calibrated
estimator state, quantizer parameters, the outputs of an estimating, a fixed-range and a train-mode forward, and -- with
learnable ranges (QAT) -- the straight-through gradients w.r.t. the input, the weights and every range parameter.  tests/test_host_logic.py runs both sides in separate processes (both packages
are called `quantization`) and demands bit equality; skipped where /root/reference is absent.

    python oracle/fuzz_models.py --impl ref|mine --n 20 --seed 0 --out /tmp/x.npz

Random axes: depth 2-4, widths 8/16/24, ReLU / GELU folded into the Linear or none, LayerNorm in between, weight bits
4/8 and activation bits 4/8/16, per-channel weights, weight estimator current min-max / MSE grid / MSE golden section,
activation estimator current / running (momentum) / all-time min-max / MSE grid, 1-3 calibration batches.  A second
family drives ONE activation site on [B, T, d] hidden states through the transformer granularities of
utils/per_embd_quant_utils.py (per-embedding, N groups, range-permuted groups incl. the phase-1 range collection, bit
width overrides, 'fp32') -- shapes of the parameter buffers included.  A third
family runs AdaRound (apply_adaround_to_layer) on both Linears of a small network: alpha, grid, reported losses, output;
a fourth the calibration driver utils.pass_data_for_range_estimation (switches, batch limits, tuple / dict batches, the
cross-entropy estimator on a named layer); a fifth the model-level drivers utils.adaround_utils.apply_adaround_to_model and
utils.qat_utils.prepare_model_for_quantization (manager states and trainable-parameter names included); a sixth
MobileBERT's QuantNoNorm with its shared weight / bias quantizer.
Test infrastructure (like everything under oracle/); needs /root/reference for --impl ref.
"""
import argparse
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(impl):
    if impl == 'ref':
        ref = '/root/reference'
        sys.path.insert(0, ref)
        u = types.ModuleType('utils')           # namespace stand-in: the reference's utils/__init__ pulls in TensorBoard
        u.__path__ = [os.path.join(ref, 'utils')]
        for name in ('_tb_advance_global_step', '_tb_advance_token_counters', '_tb_hist'):
            setattr(u, name, lambda *a, **k: None)
        sys.modules['utils'] = u
        from utils.utils import DotDict
        u.DotDict = DotDict
        import transformers.modeling_utils as mu          # transformers 4.1 name the reference's models/ import
        from transformers.pytorch_utils import apply_chunking_to_forward
        mu.apply_chunking_to_forward = apply_chunking_to_forward
    else:
        sys.path.insert(0, os.path.join(ROOT, 'transformer-quantization_amd'))
        sys.path.insert(0, ROOT)
        from quantization import _hip
        from tests._oracle_backend import OracleBackend
        _hip.set_backend(OracleBackend())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--impl', choices=['ref', 'mine'], required=True)
    ap.add_argument('--n', type=int, default=20)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--out', required=True)
    ap.add_argument('--double', action='store_true',
                    help='main.py:227-231 (--double): the first family in float64, parameter caching off (the reference\'s eval '
                         'cache narrows float64 weights to fp32 and its next forward fails); other families are skipped')
    ap.add_argument('--inplace-state', action='store_true',
                    help='(mine) options.INPLACE_CALIBRATION_STATE: calibration updates its state buffers in place')
    args = ap.parse_args()
    _setup(args.impl)
    if args.inplace_state:
        from quantization import options
        options.INPLACE_CALIBRATION_STATE = True

    import numpy as np
    import torch
    from torch import nn
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_classes import QuantizedModule
    from quantization.quantization_manager import QuantizationManager
    from quantization.quantizers import QMethods
    from quantization.range_estimators import OptMethod, RangeEstimators

    torch.set_num_threads(1)
    out = {}
    for case in range(args.n):
        rs = np.random.RandomState(1000 * args.seed + case)
        torch.manual_seed(1000 * args.seed + case)
        depth = int(rs.randint(2, 5))
        widths = [int(rs.choice([8, 16, 24])) for _ in range(depth + 1)]
        layers = []
        # usually one activation style per network: the rewriter folds the FIRST activation found anywhere after a Linear
        # into it and skips what lies between (reference autoquant_utils.py get_act / quant_module) -- a mixed network
        # mostly ends in a shape error on both sides; one case in ten keeps the free mix to cover exactly that
        style = int(rs.randint(3))
        wild = rs.randint(10) == 0
        for i in range(depth):
            layers.append(nn.Linear(widths[i], widths[i + 1], bias=bool(rs.randint(4))))
            a = rs.randint(3) if wild else style
            if a == 1:
                layers.append(nn.ReLU())
            elif a == 2:
                layers.append(nn.GELU())
            # a LayerNorm is rewritten like a Linear (the next activation ANYWHERE after it is folded into it, what lies
            # between is skipped): valid only as the last module unless the network is activation-free
            if (rs.randint(3) == 0) if (wild or style == 0) else (i == depth - 1 and rs.randint(2) == 0):
                layers.append(nn.LayerNorm(widths[i + 1]))
        # token ids -> QuantEmbedding (weight-only quantization, FP32 output site).  As an ATTRIBUTE of a container module,
        # like HF models hold theirs: inside a Sequential the reference's rewriter fails on `Embedding.bias`
        embed = rs.randint(4) == 0
        net = nn.Sequential(*layers)
        if embed:
            class WithEmbedding(nn.Module):
                def __init__(self, body):
                    super().__init__()
                    self.emb = nn.Embedding(20, widths[0])
                    self.body = body

                def forward(self, ids):
                    return self.body(self.emb(ids))
            net = WithEmbedding(net)
        w_est = [(RangeEstimators.current_minmax, None), (RangeEstimators.MSE, dict(num_candidates=10)),
                 (RangeEstimators.MSE, dict(opt_method=OptMethod.golden_section))][rs.randint(3)]
        a_est = [(RangeEstimators.current_minmax, None), (RangeEstimators.running_minmax, dict(momentum=float(rs.choice([0.9, 0.5])))),
                 (RangeEstimators.allminmax, None), (RangeEstimators.MSE, dict(num_candidates=6))][rs.randint(4)]
        n_bits_act = int(rs.choice([4, 8, 16]))
        if a_est[0] == RangeEstimators.MSE:
            n_bits_act = 4                     # 2-D grid: 6 x 4 x 2 candidates per batch
        qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform,
                  n_bits=int(rs.choice([4, 8])), n_bits_act=n_bits_act,
                  per_channel_weights=bool(rs.randint(2)),
                  weight_range_method=w_est[0], act_range_method=a_est[0])
        # rarer settings of the module kwargs (base_quantized_classes.py:36-47)
        if rs.randint(6) == 0:
            qp['method'] = QMethods.asymmetric_uniform            # asymmetric weight grid
        if rs.randint(6) == 0:
            qp['act_method'] = QMethods.symmetric_uniform         # symmetric activation grid
        if rs.randint(5) == 0:
            qp['percentile'] = float(rs.choice([99.0, 99.9, 90.0]))
        if rs.randint(8) == 0:
            qp['scale_domain'] = 'log'
        if rs.randint(8) == 0:
            qp['per_channel_acts'] = True
        if rs.randint(8) == 0:
            qp['n_bits'] = 16
        if (qp['per_channel_weights'] and w_est[1] and 'opt_method' in w_est[1] and rs.randint(5)
                and any(isinstance(l, nn.LayerNorm) for l in layers)):
            qp['per_channel_weights'] = False       # per-channel golden section on a 1-D weight raises (both sides): keep 1 in 5
        if w_est[1]:
            qp['weight_range_options'] = w_est[1]
        if a_est[1]:
            qp['act_range_options'] = a_est[1]
        tag = f'c{case}_'
        try:
            qm = quantize_model(net, **qp)
            B = int(rs.randint(2, 6))
            if embed:
                batches = [torch.tensor(rs.randint(0, 20, size=(B, 5))) for _ in range(int(rs.randint(1, 4)))]
                x_eval = torch.tensor(rs.randint(0, 20, size=(B, 5)))
            else:
                batches = [torch.tensor((rs.randn(B, widths[0]) * 10 ** rs.uniform(-1, 1)).astype(np.float32))
                           for _ in range(int(rs.randint(1, 4)))]
                x_eval = torch.tensor(rs.randn(B, widths[0]).astype(np.float32))
            for m in qm.modules():
                if isinstance(m, QuantizedModule):
                    m.quantized()
            if args.double:
                qm.double()
                for m in qm.modules():
                    if hasattr(m, '_caching'):
                        m._caching = False
                if not embed:
                    batches, x_eval = [b_.double() for b_ in batches], x_eval.double()
            qm.eval()
            with torch.no_grad():
                for k, xb in enumerate(batches):
                    out[tag + f'y_est{k}'] = qm(xb).numpy().copy()
                mgrs = [(n, m) for n, m in qm.named_modules() if isinstance(m, QuantizationManager)]
                for n, m in mgrs:
                    if m.quantizer.is_initialized:
                        m.fix_ranges()
                out[tag + 'y_fixed'] = qm(x_eval).numpy().copy()
            # checkpoint layout: key names, shapes and dtypes of the calibrated model's state_dict
            out[tag + 'state_dict'] = np.array(sorted(f'{k}:{tuple(v.shape)}:{v.dtype}' for k, v in qm.state_dict().items()))
            qm.train()                              # train mode: no weight cache, ranges stay fixed
            out[tag + 'y_train'] = qm(x_eval).detach().numpy().copy()
            # learnable ranges (QAT): straight-through backward w.r.t. input, weights and every range parameter
            for n, m in mgrs:
                if m.quantizer.is_initialized:
                    m.learn_ranges()
            xg = x_eval.clone() if embed else x_eval.clone().requires_grad_(True)
            yq = qm(xg)
            gy = torch.tensor(rs.randn(*yq.shape).astype(np.float32)).to(yq.dtype)
            (yq * gy).sum().backward()
            out[tag + 'y_learn'] = yq.detach().numpy().copy()
            if not embed:
                out[tag + 'gx'] = xg.grad.numpy().copy()
            for pn, prm in qm.named_parameters():
                if prm.grad is not None:
                    out[tag + 'grad.' + pn] = prm.grad.numpy().reshape(-1).copy()
            for n, m in mgrs:
                q = m.quantizer
                if not q.is_initialized:
                    continue
                out[tag + n + '.delta'] = q._delta.detach().numpy().reshape(-1).copy()
                zf = getattr(q, '_zero_float', None)
                if zf is not None:
                    out[tag + n + '.zero_float'] = zf.detach().numpy().reshape(-1).copy()
                sg = getattr(q, '_signed', None)
                if sg is not None:
                    out[tag + n + '.signed'] = np.array(bool(sg))
                est = m.range_estimator
                if est is not None and getattr(est, 'current_xmin', None) is not None:
                    out[tag + n + '.xmin'] = torch.as_tensor(est.current_xmin).detach().numpy().reshape(-1).copy()
                    out[tag + n + '.xmax'] = torch.as_tensor(est.current_xmax).detach().numpy().reshape(-1).copy()
        except Exception as e:      # error behaviour is part of the contract: same exception type on both sides
            for k in [k for k in out if k.startswith(tag)]:
                del out[k]
            out[tag + 'raised'] = np.array(type(e).__name__)
        out[tag + 'cfg'] = np.array(repr({k: str(v) for k, v in qp.items()}) + f' depth={depth} widths={widths}')

    if args.double:
        np.savez_compressed(args.out, **out)
        print('cases', args.n, 'arrays', len(out), '(float64, first family only)')
        return

    # ---- second family: one activation site on [B, T, d] hidden states with the transformer-specific granularities
    # (utils/per_embd_quant_utils.py: per-embedding, N groups, range-permuted groups, quant_dict letter-code values)
    from quantization.base_quantized_classes import QuantizedActivation
    from utils.per_embd_quant_utils import hijack_act_quant
    for case in range(args.n):
        rs = np.random.RandomState(77000 + 1000 * args.seed + case)
        tag = f's{case}_'
        d = int(rs.choice([12, 24, 48]))
        B, T = int(rs.randint(1, 4)), int(rs.randint(2, 7))
        code = [None, 'per_embd', 'ng2', 'ng6', 'ngp3', 'ngp6', int(rs.choice([4, 6, 16])), 'fp32'][rs.randint(8)]
        est = [(RangeEstimators.current_minmax, None), (RangeEstimators.running_minmax, dict(momentum=0.9)),
               (RangeEstimators.allminmax, None), (RangeEstimators.MSE, dict(num_candidates=5))][rs.randint(4)]
        if isinstance(code, str) and code.startswith('ngp') and rs.randint(4):
            est = (RangeEstimators.current_minmax, None)     # the only estimator with a phase 1 (range collection)
        qp = dict(act_method=QMethods.asymmetric_uniform, n_bits_act=int(rs.choice([4, 8])), act_range_method=est[0])
        if est[1]:
            qp['act_range_options'] = est[1]
        xs = [torch.tensor((rs.randn(B, T, d) * 10 ** rs.uniform(-1, 1)).astype(np.float32)) for _ in range(int(rs.randint(1, 4)))]
        for x in xs:                        # a few outlier dimensions, as in BERT's residual stream
            x[..., rs.randint(d)] *= 20.0
        try:
            site = QuantizedActivation(**qp)
            hijack_act_quant({'y': code}, 'y', site)
            site.quantized_acts()
            site.eval()
            mgr = site.activation_quantizer
            with torch.no_grad():
                estr = getattr(mgr, 'range_estimator', None)
                if estr is not None and getattr(estr, 'per_group_range_estimation', False):
                    out[tag + 'y_phase1'] = site(xs[0]).numpy().copy()        # phase 1: passthrough, collects ranges
                    out[tag + 'ranges'] = estr.ranges.numpy().reshape(-1).copy()
                    estr.per_group_range_estimation = False
                for k, x in enumerate(xs):
                    out[tag + f'y_est{k}'] = site(x).numpy().copy()
                if hasattr(mgr, 'fix_ranges'):
                    mgr.fix_ranges()
                out[tag + 'y_fixed'] = site(xs[-1] * 1.5).numpy().copy()
            out[tag + 'state_dict'] = np.array(sorted(f'{k}:{tuple(v.shape)}:{v.dtype}' for k, v in site.state_dict().items()))
            if hasattr(mgr, 'quantizer'):
                out[tag + 'delta'] = mgr.quantizer._delta.detach().numpy().copy()
                out[tag + 'zero_float'] = mgr.quantizer._zero_float.detach().numpy().copy()
                out[tag + 'xmin'] = torch.as_tensor(mgr.range_estimator.current_xmin).numpy().copy()
                out[tag + 'xmax'] = torch.as_tensor(mgr.range_estimator.current_xmax).numpy().copy()
        except Exception as e:
            for k in [k for k in out if k.startswith(tag)]:
                del out[k]
            out[tag + 'raised'] = np.array(type(e).__name__)
        out[tag + 'cfg'] = np.array(f'code={code} est={est[0]} d={d} B={B} T={T} bits={qp["n_bits_act"]}')

    # ---- third family: AdaRound of both Linears of a small network through apply_adaround_to_layer (grid inits
    # range_estimator / mse / mse_out / mse_out_asym, symmetric or asymmetric input-caching, with / without the activation function,
    # learned_sigmoid / learned_hard_sigmoid / sigmoid_temp_decay, 15-30 iterations; mini-batch indices come from the global torch RNG on
    # both sides, reference adaround/adaround.py:236): alpha, grid, the four reported losses and the final output
    from quantization.adaround import apply_adaround_to_layer
    from quantization.adaround.config import DEFAULT_ADAROUND_CONFIG
    from quantization.adaround.utils import AdaRoundInitMode, AdaRoundMode
    from quantization.base_quantized_model import QuantizedModel
    from utils.utils import DotDict
    for case in range(max(args.n // 4, 1)):
        rs = np.random.RandomState(55000 + 1000 * args.seed + case)
        torch.manual_seed(55000 + 1000 * args.seed + case)
        d0, d1, d2 = [int(rs.choice([8, 16])) for _ in range(3)]
        qp = dict(method=QMethods.symmetric_uniform if rs.randint(2) else QMethods.asymmetric_uniform,
                  act_method=QMethods.asymmetric_uniform, n_bits=int(rs.choice([3, 4])), n_bits_act=8,
                  weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)

        class Toy(QuantizedModel):
            def __init__(self):
                super().__init__()
                self.body = quantize_model(nn.Sequential(nn.Linear(d0, d1), nn.ReLU(), nn.Linear(d1, d2)), **qp)

            def forward(self, x):
                return self.body(x)

        tag = f'a{case}_'
        cfg = DotDict(dict(DEFAULT_ADAROUND_CONFIG))
        cfg.iters, cfg.lr = int(rs.choice([15, 30])), 1e-2
        cfg.round_mode = [AdaRoundMode.learned_hard_sigmoid, AdaRoundMode.learned_sigmoid,
                          AdaRoundMode.sigmoid_temp_decay][rs.randint(3)]
        cfg.init = [AdaRoundInitMode.range_estimator, AdaRoundInitMode.mse, AdaRoundInitMode.mse_out,
                    AdaRoundInitMode.mse_out_asym][rs.randint(4)]
        cfg.asym, cfg.include_act_func = bool(rs.randint(2)), bool(rs.randint(2))
        try:
            model = Toy()
            data = torch.tensor(rs.randn(48, d0).astype(np.float32))
            model.eval()
            model.set_quant_state(True, False)
            with torch.no_grad():
                model(data[:16])                  # weight ranges
            model.fix_ranges()
            for li, layer in enumerate([model.body[0], model.body[1]]):
                torch.manual_seed(9000 + case * 10 + li)
                res = apply_adaround_to_layer(model, layer, data, batch_size=8, act_quant=False, adaround_config=cfg,
                                              keep_gpu=False)
                out[tag + f'L{li}_losses'] = np.array([float(res.loss_soft_before), float(res.loss_hard_before),
                                                       float(res.loss_soft_after), float(res.loss_hard_after)])
                out[tag + f'L{li}_alpha'] = layer.weight_quantizer.quantizer.alpha.detach().numpy().copy()
                out[tag + f'L{li}_delta'] = layer.weight_quantizer.quantizer._delta.detach().numpy().reshape(-1).copy()
            with torch.no_grad():
                out[tag + 'y'] = model(data[:8]).numpy().copy()
        except Exception as e:
            for k in [k for k in out if k.startswith(tag)]:
                del out[k]
            out[tag + 'raised'] = np.array(type(e).__name__)
        out[tag + 'cfg'] = np.array(f'{cfg.round_mode} {cfg.init} asym={cfg.asym} act={cfg.include_act_func} iters={cfg.iters}')

    # ---- fourth family: the calibration driver utils.pass_data_for_range_estimation on a QuantizedModel -- weight_quant /
    # act_quant switches, max_num_batches, tuple and dict batches, the cross-entropy estimator installed on a named layer
    # (reference utils/utils.py:47-79) -- followed by model.fix_ranges() and a forward
    from utils.utils import pass_data_for_range_estimation
    for case in range(max(args.n // 2, 1)):
        rs = np.random.RandomState(33000 + 1000 * args.seed + case)
        torch.manual_seed(33000 + 1000 * args.seed + case)
        d0, d1, d2 = int(rs.choice([8, 16])), int(rs.choice([8, 16])), int(rs.choice([2, 3, 5]))
        a_est = [(RangeEstimators.current_minmax, None), (RangeEstimators.running_minmax, None),
                 (RangeEstimators.MSE, dict(num_candidates=6))][rs.randint(3)]
        qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=int(rs.choice([4, 8])),
                  n_bits_act=4 if a_est[0] == RangeEstimators.MSE else 8, per_channel_weights=bool(rs.randint(2)),
                  weight_range_method=RangeEstimators.current_minmax, act_range_method=a_est[0])
        if a_est[1]:
            qp['act_range_options'] = a_est[1]

        class Head(QuantizedModel):
            def __init__(self):
                super().__init__()
                self.body = quantize_model(nn.Sequential(nn.Linear(d0, d1), nn.ReLU(), nn.Linear(d1, d2)), **qp)

            def forward(self, x=None, scale=None):
                return self.body(x if scale is None else x * scale)

        tag = f'p{case}_'
        act_quant, weight_quant = bool(rs.randint(4)), bool(rs.randint(4))
        xent = [None, 'body.1', 'body.7'][rs.randint(3)] if rs.randint(2) else None      # 'body.7': no such layer
        n_batches, max_b = int(rs.randint(1, 5)), int(rs.randint(1, 4))
        as_dict = bool(rs.randint(2))
        try:
            model = Head()
            xs = [torch.tensor((rs.randn(4, d0) * (1 + k)).astype(np.float32)) for k in range(n_batches)]
            loader = [dict(x=x, scale=torch.tensor(1.5)) for x in xs] if as_dict else [(torch.zeros(1), x) for x in xs]
            kw = {} if as_dict else dict(inp_idx=1)
            with torch.no_grad():
                pass_data_for_range_estimation(loader, model, act_quant, weight_quant, max_num_batches=max_b,
                                               cross_entropy_layer=xent, **kw)
                if rs.randint(3) == 0:            # start over: reset the activation ranges, one more calibration batch
                    model.reset_act_ranges()
                    pass_data_for_range_estimation(loader[-1:], model, act_quant, weight_quant, max_num_batches=1, **kw)
                model.fix_ranges()
                out[tag + 'y'] = model(xs[0]).numpy().copy()
            for n, m in model.named_modules():
                if isinstance(m, QuantizationManager) and m.quantizer.is_initialized:
                    out[tag + n + '.delta'] = m.quantizer._delta.detach().numpy().reshape(-1).copy()
                    zf = getattr(m.quantizer, '_zero_float', None)
                    if zf is not None:
                        out[tag + n + '.zero_float'] = zf.detach().numpy().reshape(-1).copy()
                    out[tag + n + '.estimator'] = np.array(type(m.range_estimator).__name__)
        except Exception as e:
            for k in [k for k in out if k.startswith(tag)]:
                del out[k]
            out[tag + 'raised'] = np.array(type(e).__name__)
        out[tag + 'cfg'] = np.array(f'act={act_quant} w={weight_quant} xent={xent} batches={n_batches}/{max_b} dict={as_dict} est={a_est[0]}')

    # ---- fifth family: the model-level drivers -- utils.adaround_utils.apply_adaround_to_model (layer selection 'all' /
    # by name, act_quant_mode post_adaround / no_act_quant with the re-calibration of activation ranges that follows)
    # and utils.qat_utils.prepare_model_for_quantization (learn_ranges / estimate-in-train / fixed weight and activation
    # ranges), each followed by a forward
    from quantization.adaround.utils import AdaRoundActQuantMode
    from utils.adaround_utils import apply_adaround_to_model
    from utils.qat_utils import prepare_model_for_quantization
    for case in range(max(args.n // 4, 1)):
        rs = np.random.RandomState(21000 + 1000 * args.seed + case)
        torch.manual_seed(21000 + 1000 * args.seed + case)
        d0, d1, d2 = int(rs.choice([8, 16])), int(rs.choice([8, 16])), int(rs.choice([2, 4]))
        qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=4, n_bits_act=8,
                  weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)

        class Net(QuantizedModel):
            def __init__(self):
                super().__init__()
                self.body = quantize_model(nn.Sequential(nn.Linear(d0, d1), nn.ReLU(), nn.Linear(d1, d2)), **qp)

            def forward(self, x):
                return self.body(x)

        tag = f'm{case}_'
        which = 'adaround' if rs.randint(2) else 'qat'
        cfg = DotDict(quant=DotDict(act_quant=bool(rs.randint(3)), weight_quant=True),
                      act_quant=DotDict(num_batches=int(rs.randint(1, 3)), cross_entropy_layer=None),
                      qat=DotDict(learn_ranges=bool(rs.randint(2)), fix_weight_ranges=bool(rs.randint(2)),
                                  fix_act_ranges=bool(rs.randint(2))))
        ada = DotDict(dict(DEFAULT_ADAROUND_CONFIG))
        ada.iters, ada.lr, ada.num_samples = 12, 1e-2, 24
        ada.layers = [('all',), ('body.0',), ('body.1', 'body.0')][rs.randint(3)]
        ada.act_quant_mode = [AdaRoundActQuantMode.post_adaround, AdaRoundActQuantMode.no_act_quant][rs.randint(2)]
        cfg.adaround = ada
        try:
            model = Net()
            xs = [torch.tensor(rs.randn(8, d0).astype(np.float32)) for _ in range(4)]
            loader = [(x, torch.zeros(8)) for x in xs]
            if which == 'adaround':
                with torch.no_grad():
                    pass_data_for_range_estimation(loader, model, act_quant=False, weight_quant=True, max_num_batches=1)
                model.fix_ranges()
                torch.manual_seed(4242 + case)
                apply_adaround_to_model(cfg, model, loader, loader, batch_size=8)
            else:
                prepare_model_for_quantization(cfg, model, loader)
                model.train()
            y = model(xs[0])
            out[tag + 'y'] = y.detach().numpy().copy()
            for n, m in model.named_modules():
                if isinstance(m, QuantizationManager):
                    out[tag + n + '.state'] = np.array(str(m.state))
                    if m.quantizer.is_initialized:
                        out[tag + n + '.delta'] = m.quantizer._delta.detach().numpy().reshape(-1).copy()
                        al = getattr(m.quantizer, 'alpha', None)
                        if al is not None:
                            out[tag + n + '.alpha'] = al.detach().numpy().copy()
            out[tag + 'trainable'] = np.array(sorted(n for n, p_ in model.named_parameters() if p_.requires_grad))
            out[tag + 'state_dict'] = np.array(sorted(f'{k}:{tuple(v.shape)}:{v.dtype}' for k, v in model.state_dict().items()))
        except Exception as e:
            for k in [k for k in out if k.startswith(tag)]:
                del out[k]
            out[tag + 'raised'] = np.array(type(e).__name__)
        out[tag + 'cfg'] = np.array(f'{which} layers={ada.layers} mode={ada.act_quant_mode} act_quant={cfg.quant.act_quant} qat={dict(cfg.qat)}')

    # ---- sixth family: MobileBERT's QuantNoNorm (reference models/quantized_mobilebert.py:58-72; here
    # quantization.autoquant_utils.QuantNoNorm): ONE weight quantizer applied to the weight, then to the bias (while
    # estimating, the bias call overwrites the weight's range -- quirk q9), output quantizer, 1-3 calibration batches
    if args.impl == 'ref':
        from models.quantized_mobilebert import QuantNoNorm
    else:
        from quantization.autoquant_utils import QuantNoNorm
    from transformers.models.mobilebert.modeling_mobilebert import NoNorm
    for case in range(max(args.n // 2, 1)):
        rs = np.random.RandomState(66000 + 1000 * args.seed + case)
        tag = f'n{case}_'
        d = int(rs.choice([8, 16, 128]))
        w_est = [(RangeEstimators.current_minmax, None), (RangeEstimators.MSE, dict(num_candidates=8))][rs.randint(2)]
        qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=int(rs.choice([4, 8])),
                  n_bits_act=int(rs.choice([4, 8])), weight_range_method=w_est[0],
                  act_range_method=[RangeEstimators.running_minmax, RangeEstimators.current_minmax][rs.randint(2)])
        if w_est[1]:
            qp['weight_range_options'] = w_est[1]
        try:
            org = NoNorm(d)
            org.weight.data = torch.tensor((1.0 + 0.3 * rs.randn(d)).astype(np.float32))
            org.bias.data = torch.tensor((0.2 * rs.randn(d)).astype(np.float32))
            m = QuantNoNorm(org, **qp)
            m.quantized()
            m.eval()
            xs = [torch.tensor((rs.randn(2, 5, d) * (1 + 0.5 * k)).astype(np.float32)) for k in range(int(rs.randint(1, 4)))]
            with torch.no_grad():
                for k, x in enumerate(xs):
                    out[tag + f'y_est{k}'] = m(x).numpy().copy()
                m.weight_quantizer.fix_ranges()
                m.activation_quantizer.fix_ranges()
                out[tag + 'y_fixed'] = m(xs[0] * 0.7).numpy().copy()
            out[tag + 'w_delta'] = m.weight_quantizer.quantizer._delta.numpy().reshape(-1).copy()
            out[tag + 'a_delta'] = m.activation_quantizer.quantizer._delta.numpy().reshape(-1).copy()
            out[tag + 'a_zero_float'] = m.activation_quantizer.quantizer._zero_float.numpy().reshape(-1).copy()
            out[tag + 'state_dict'] = np.array(sorted(f'{k}:{tuple(v.shape)}:{v.dtype}' for k, v in m.state_dict().items()))
        except Exception as e:
            for k in [k for k in out if k.startswith(tag)]:
                del out[k]
            out[tag + 'raised'] = np.array(type(e).__name__)
        out[tag + 'cfg'] = np.array(f'd={d} {qp}')

    np.savez_compressed(args.out, **out)
    print('cases', args.n, 'arrays', len(out))


if __name__ == '__main__':
    main()
