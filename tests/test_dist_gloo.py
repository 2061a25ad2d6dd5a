"""N > 1 path on CPU: world_size-2 `gloo` process groups (the same torch.distributed calls go to
RCCL on the GPU box).  The backend is the oracle-backed double, so what is verified here is the
sharding logic: per-rank statistics all-reduced inline give every rank the statistics of the
concatenated batch (exact for min/max; equal argmin for the MSE search), and data-parallel AdaRound
reproduces the single-rank step on the same global batch.
"""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import ROOT, PKG

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup(rank, port):
    for p in (PKG, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    from quantization import _hip, distributed as tq_dist
    from tests._oracle_backend import OracleBackend
    _hip.set_backend(OracleBackend())
    tq_dist.enable()
    return tq_dist


def _batches(n=3, B=4, T=6, D=24, seed=0):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        x = torch.randn(B, T, D, generator=g) * (1.0 + 0.5 * i)
        x[..., 3] *= 15
        out.append(x)
    return out


def _run_managers(xs, shard):
    """Drive a set of managers over batches; `shard(x)` picks the local slice."""
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from quantization.quantization_manager import QuantizationManager
    from utils.per_embd_quant_utils import set_act_quant_axis_and_groups
    cfgs = [('asymmetric_uniform', 'running_minmax', None, {}),
            ('symmetric_uniform', 'allminmax', None, {}),
            ('asymmetric_uniform', 'current_minmax', 'per_embd', {}),
            ('asymmetric_uniform', 'running_minmax', 'peg4', {}),
            ('symmetric_uniform', 'MSE', None, dict(num_candidates=40)),
            ('asymmetric_uniform', 'MSE', None, dict(num_candidates=8)),
            ('asymmetric_uniform', 'cross_entropy', 'logits', dict(num_candidates=16))]
    res = []
    for method, init, layout, ip in cfgs:
        mgr = QuantizationManager(qmethod=QMethods[method], init=RangeEstimators[init],
                                  qparams=dict(n_bits=4 if init == 'MSE' else 8), init_params=ip)
        if layout == 'per_embd':
            set_act_quant_axis_and_groups(mgr, axis=2, n_groups=None)
        elif layout == 'peg4':
            set_act_quant_axis_and_groups(mgr, axis=2, n_groups=4)
        for x in xs:
            # cross-entropy estimator: [B, num_labels] logits (rows shard cleanly: softmax is per row)
            mgr(shard(x[:, 0, :3].contiguous() if layout == 'logits' else x))
        est = mgr.range_estimator
        rec = dict(xmin=est.current_xmin.clone(), xmax=est.current_xmax.clone(),
                   delta=mgr.quantizer._delta.clone())
        if init in ('MSE', 'cross_entropy'):
            rec['loss'] = est.loss_array.copy()
        res.append(rec)
    return res


def _worker_calibration(rank, port, outdir):
    tq_dist = _setup(rank, port)
    xs = _batches()
    res = _run_managers(xs, tq_dist.shard_batch)
    torch.save(res, os.path.join(outdir, f'calib_{rank}.pt'))
    torch.save(tq_dist.stats(), os.path.join(outdir, f'stats_{rank}.pt'))
    # ragged shard: 5 samples over 2 ranks -> 3 + 2
    x = torch.arange(5.0).view(5, 1)
    assert tq_dist.shard_batch(x).shape[0] == (3 if rank == 0 else 2)
    dist.destroy_process_group()


def test_sharded_calibration_equals_single_rank(tmp_path):
    port = _free_port()
    mp.spawn(_worker_calibration, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    # single-rank reference on the full batches (same backend double, collectives off)
    from quantization import _hip, distributed as tq_dist
    from tests._oracle_backend import OracleBackend
    prev = _hip.set_backend(OracleBackend())
    try:
        tq_dist.disable()
        ref = _run_managers(_batches(), lambda x: x)
    finally:
        _hip.set_backend(prev)
    for rank in range(WORLD):
        got = torch.load(os.path.join(tmp_path, f'calib_{rank}.pt'), weights_only=False)
        for i, (g, r) in enumerate(zip(got, ref)):
            assert torch.equal(g['xmin'], r['xmin']), (rank, i)      # min/max: exact; searches: same argmin
            assert torch.equal(g['xmax'], r['xmax']), (rank, i)
            assert torch.equal(g['delta'], r['delta']), (rank, i)
            if 'loss' in r:
                fin = np.isfinite(r['loss'])
                assert np.allclose(g['loss'][fin], r['loss'][fin], rtol=1e-6), (rank, i)
        st = torch.load(os.path.join(tmp_path, f'stats_{rank}.pt'), weights_only=False)
        assert st['minmax_calls'] > 0 and st['sum_calls'] > 0


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = torch.nn.Linear(24, 32)
        self.fc2 = torch.nn.Linear(32, 24)


def _quant_toy(org):
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from quantization.base_quantized_model import QuantizedModel
    from quantization.base_quantized_classes import QuantizedActivation
    from quantization.autoquant_utils import quantize_model
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8,
              act_range_method=RangeEstimators.running_minmax)

    class Q(QuantizedModel):
        def __init__(self):
            super().__init__()
            self.fc1 = quantize_model(org.fc1, **qp)
            self.fc2 = quantize_model(org.fc2, **qp)
            self.res = QuantizedActivation(**qp)

        def forward(self, x):
            return self.res(self.fc2(torch.relu(self.fc1(x))) + x)
    return Q()


def _calibrated_state(shard_loader):
    from utils.utils import pass_data_for_range_estimation
    torch.manual_seed(5)
    model = _quant_toy(_Toy())
    loader = [(b,) for b in _batches(n=3, B=8)]
    pass_data_for_range_estimation(loader, model, act_quant=True, weight_quant=True, max_num_batches=3)
    model.fix_ranges()
    return {k: v.clone() for k, v in model.state_dict().items()
            if any(s in k for s in ('_delta', '_zero_float', 'current_x'))}


def _worker_model(rank, port, outdir):
    _setup(rank, port)
    torch.save(_calibrated_state(None), os.path.join(outdir, f'model_{rank}.pt'))
    dist.destroy_process_group()


def test_sharded_pass_data_for_range_estimation(tmp_path):
    """The calibration driver shards every batch along dim 0; the resulting quantizer state is the
    single-rank state (activation ranges within fp32 GEMM round-off of the batch split, weight
    ranges exactly)."""
    port = _free_port()
    mp.spawn(_worker_model, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    from quantization import _hip, distributed as tq_dist
    from tests._oracle_backend import OracleBackend
    prev = _hip.set_backend(OracleBackend())
    try:
        tq_dist.disable()
        ref = _calibrated_state(None)
    finally:
        _hip.set_backend(prev)
    for rank in range(WORLD):
        got = torch.load(os.path.join(tmp_path, f'model_{rank}.pt'), weights_only=False)
        assert got.keys() == ref.keys()
        for k in ref:
            if 'weight_quantizer' in k:
                assert torch.equal(got[k], ref[k]), k
            else:
                assert torch.allclose(got[k], ref[k], rtol=1e-5, atol=1e-6), k


def _ada_problem(device='cpu'):
    from quantization.quantizers import QMethods
    from quantization.autoquant_utils import QuantLinear
    from quantization.adaround.quantizer import ADAROUND_QUANTIZER_MAP
    from quantization.adaround.utils import AdaRoundMode, CombinedLoss, AdaRoundLossType
    from quantization.adaround.adaround import FusedAlphaAdam
    g = torch.Generator().manual_seed(31)
    layer = QuantLinear(16, 12, method=QMethods.symmetric_uniform, n_bits=4)
    layer.weight.data = torch.randn(12, 16, generator=g) * 0.3
    layer.bias.data = torch.randn(12, generator=g) * 0.1
    layer.quantized_weights()
    layer.caching = False
    layer.to(device)
    X = torch.randn(16, 5, 16, generator=g).to(device)
    tgt = torch.randn(16, 5, 12, generator=g).to(device)
    with torch.no_grad():
        layer(X)
    oq = layer.weight_quantizer.quantizer
    wq = ADAROUND_QUANTIZER_MAP[oq.__class__](n_bits=4)
    for name in ('_delta', '_zero_float', '_signed'):
        wq.register_buffer(name, getattr(oq, name))
    layer.weight_quantizer.quantizer = wq
    layer.weight_quantizer.fix_ranges()
    wq.round_mode = AdaRoundMode.learned_hard_sigmoid
    wq.soft_targets = True
    with torch.no_grad():
        wq(layer.weight)
    loss_fn = CombinedLoss(quantizer=wq, loss_type=AdaRoundLossType.relaxation, weight=0.01,
                           max_count=6, b_range=(20, 2), warmup=0.2)
    idx = [torch.randperm(16, generator=g)[:8].numpy() for _ in range(6)]

    def get_inp_out(data):
        pos = [int((X == d).all(-1).all(-1).nonzero()[0]) for d in data]
        return data, tgt[pos]
    return layer, wq, X, get_inp_out, loss_fn, FusedAlphaAdam(wq, lr=1e-2), idx


def _worker_adaround(rank, port, outdir):
    _setup(rank, port)
    from quantization.adaround.adaround import optimize_local_loss
    layer, wq, X, gio, loss_fn, opt, idx = _ada_problem()
    optimize_local_loss(layer, gio, X, opt, loss_fn, 8, 6, batch_indices=idx)
    torch.save(wq.alpha.detach().clone(), os.path.join(outdir, f'alpha_{rank}.pt'))
    dist.destroy_process_group()


def test_data_parallel_adaround_equals_single_rank(tmp_path):
    port = _free_port()
    mp.spawn(_worker_adaround, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    from quantization import _hip, distributed as tq_dist
    from quantization.adaround.adaround import optimize_local_loss
    from tests._oracle_backend import OracleBackend
    prev = _hip.set_backend(OracleBackend())
    try:
        tq_dist.disable()
        layer, wq, X, gio, loss_fn, opt, idx = _ada_problem()
        optimize_local_loss(layer, gio, X, opt, loss_fn, 8, 6, batch_indices=idx)
        ref = wq.alpha.detach().clone()
    finally:
        _hip.set_backend(prev)
    a0 = torch.load(os.path.join(tmp_path, 'alpha_0.pt'), weights_only=False)
    a1 = torch.load(os.path.join(tmp_path, 'alpha_1.pt'), weights_only=False)
    assert torch.equal(a0, a1)                                    # replicas stay in lock-step
    assert torch.allclose(a0, ref, rtol=1e-4, atol=1e-5)          # and follow the 1-rank trajectory


def test_bench_gpus_flag_spawns_ranks():
    """`python bench.py --gpus 2` (no torchrun environment) re-executes itself under torch.distributed.run with 2
    ranks.  TQ_BENCH_DRY_RUN=1 keeps it to the launcher / rendezvous / all-reduce / rank-0 reporting control flow
    (gloo, no kernels): this box has no GPU, and the product path has no CPU fallback to run instead."""
    import json
    import subprocess
    env = dict(os.environ, TQ_BENCH_DRY_RUN='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                                           # rank 0 only
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['rccl_world_size'] == 2 and out['dry_run'] is True and out['value'] is None
    assert out['bring_up_agreement_round'] is True                  # rccl.agree over the gloo rendezvous store, 2 ranks
    for k in ('calibration', 'config_shape', 'calibration_model', 'adaround_dp', 'qat_dp'):
        assert k in out                                             # the keys of the real line (round 4)


# ---- layer-parallel AdaRound (asym=False: independent per-layer problems, layers shard over ranks) -------------------
def _layer_parallel_problem():
    import copy
    from quantization.adaround.config import DEFAULT_ADAROUND_CONFIG
    from quantization.adaround.utils import AdaRoundInitMode, AdaRoundMode
    from quantization.quantization_manager import QuantizationManager
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from tests.test_host_logic import ToyNet, _quant_toy
    from utils.utils import DotDict
    torch.manual_seed(4242)
    org = ToyNet()
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=4, n_bits_act=8,
              weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)
    m = _quant_toy(org, **qp)
    m.eval()
    g = torch.Generator().manual_seed(7)
    X = torch.randn(16, 5, 24, generator=g)
    m.set_quant_state(weight_quant=True, act_quant=False)
    with torch.no_grad():
        m(X[:8])
    for mod in m.modules():
        if isinstance(mod, QuantizationManager) and mod.quantizer.is_initialized:
            mod.fix_ranges()
    cfg = copy.deepcopy(DEFAULT_ADAROUND_CONFIG)
    cfg.layers, cfg.num_samples, cfg.iters = ('fc1', 'fc2'), 16, 12
    cfg.asym = False
    cfg.init = AdaRoundInitMode.mse
    cfg.round_mode = AdaRoundMode.learned_hard_sigmoid
    from quantization.adaround.utils import AdaRoundActQuantMode
    cfg.act_quant_mode = AdaRoundActQuantMode.no_act_quant
    config = DotDict(adaround=cfg, quant=DotDict(act_quant=False, weight_quant=True),
                     act_quant=DotDict(num_batches=1, cross_entropy_layer=None))
    loader = [(X[i:i + 8],) for i in (0, 8)]
    return m, config, loader


def _run_layer_parallel(m, config, loader):
    from utils.adaround_utils import apply_adaround_to_model
    torch.manual_seed(99)                          # the loops draw their sample indices from the global RNG
    res = apply_adaround_to_model(config, m, loader, loader, batch_size=8)
    state = {n: (mod.weight_quantizer.quantizer.alpha.detach().clone(),
                 mod.weight_quantizer.quantizer._delta.detach().clone())
             for n, mod in m.named_modules()
             if n in ('fc1', 'fc2') and hasattr(mod.weight_quantizer.quantizer, 'alpha')}
    return res, state


def _worker_layer_parallel(rank, port, outdir):
    tq = _setup(rank, port)
    m, config, loader = _layer_parallel_problem()
    res, state = _run_layer_parallel(m, config, loader)
    assert tq.is_enabled()                         # suspended() restored the exchange
    with torch.no_grad():
        y = m(loader[0][0])
    torch.save({'res': {k: dict(v) for k, v in res.items()}, 'state': state, 'y': y}, os.path.join(outdir, f'lp_{rank}.pt'))
    dist.destroy_process_group()


def test_layer_parallel_adaround_equals_single_process(tmp_path):
    """asym=False: rank 0 learns fc1, rank 1 learns fc2 (full sample set each, no exchange inside the loops), the learned
    alphas / ranges / losses are broadcast -- both ranks end with exactly the model a single process produces when the
    per-layer RNG streams coincide (each layer's loop is the first consumer of the seed on its owner; the
    single-process reference re-seeds before each layer accordingly)."""
    port = _free_port()
    mp.spawn(_worker_layer_parallel, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    from quantization import _hip, distributed as tq_dist
    from tests._oracle_backend import OracleBackend
    prev = _hip.set_backend(OracleBackend())
    try:
        tq_dist.disable()
        ref_state = {}
        for only in ('fc1', 'fc2'):                # one layer per run = what each owner did (same seed, same order)
            m, config, loader = _layer_parallel_problem()
            config.adaround.layers = (only,)
            _, st = _run_layer_parallel(m, config, loader)
            ref_state[only] = st[only]
    finally:
        _hip.set_backend(prev)
    r0 = torch.load(os.path.join(tmp_path, 'lp_0.pt'), weights_only=False)
    r1 = torch.load(os.path.join(tmp_path, 'lp_1.pt'), weights_only=False)
    assert set(r0['res']) == set(r1['res']) == {'fc1', 'fc2'} and r0['res'] == r1['res']
    assert torch.equal(r0['y'], r1['y'])
    for n in ('fc1', 'fc2'):
        for k in (0, 1):
            assert torch.equal(r0['state'][n][k], r1['state'][n][k]), n
            assert torch.equal(r0['state'][n][k], ref_state[n][k]), n


# ---- calibration batch smaller than the world: some ranks own an EMPTY shard -------------------------------------------
def _worker_empty_shard(rank, port, outdir):
    tq_dist = _setup(rank, port)
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from quantization.quantization_manager import QuantizationManager
    from utils.per_embd_quant_utils import set_act_quant_axis_and_groups
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(1, 6, 24, generator=g) * (1 + i) for i in range(2)]        # ONE sample per batch, world 2
    out = []
    for method, init, layout, ip in (('asymmetric_uniform', 'running_minmax', None, {}),
                                     ('symmetric_uniform', 'current_minmax', None, {}),
                                     ('asymmetric_uniform', 'current_minmax', 'per_embd', {}),
                                     ('asymmetric_uniform', 'MSE', None, dict(num_candidates=8)),
                                     ('asymmetric_uniform', 'cross_entropy', 'logits', dict(num_candidates=8))):
        mgr = QuantizationManager(qmethod=QMethods[method], init=RangeEstimators[init],
                                  qparams=dict(n_bits=4 if init == 'MSE' else 8), init_params=ip)
        if layout == 'per_embd':
            set_act_quant_axis_and_groups(mgr, axis=2, n_groups=None)
        for x in xs:
            if layout == 'logits':
                x = x[:, 0, :3].contiguous()
            local = tq_dist.shard_batch(x)
            assert local.shape[0] == (1 if rank == 0 else 0)
            y = mgr(local)
            assert y.shape == local.shape
        out.append((mgr.range_estimator.current_xmin.clone(), mgr.range_estimator.current_xmax.clone(),
                    mgr.quantizer._delta.clone()))
    torch.save((out, xs), os.path.join(outdir, f'empty_{rank}.pt'))
    dist.destroy_process_group()


def test_empty_shards_contribute_the_identity(tmp_path):
    """A 1-sample calibration batch over 2 ranks (the README recipe's --est-ranges-batch-size 1): rank 1 owns nothing,
    contributes (+inf, -inf) to the exchanged statistics and zero to the candidate losses, and ends with exactly the
    ranges rank 0 -- and a single process -- finds."""
    port = _free_port()
    mp.spawn(_worker_empty_shard, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    r0, xs = torch.load(os.path.join(tmp_path, 'empty_0.pt'), weights_only=False)
    r1, _ = torch.load(os.path.join(tmp_path, 'empty_1.pt'), weights_only=False)
    from quantization import _hip, distributed as tq_dist
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from quantization.quantization_manager import QuantizationManager
    from utils.per_embd_quant_utils import set_act_quant_axis_and_groups
    from tests._oracle_backend import OracleBackend
    prev = _hip.set_backend(OracleBackend())
    try:
        tq_dist.disable()
        ref = []
        for method, init, layout, ip in (('asymmetric_uniform', 'running_minmax', None, {}),
                                         ('symmetric_uniform', 'current_minmax', None, {}),
                                         ('asymmetric_uniform', 'current_minmax', 'per_embd', {}),
                                         ('asymmetric_uniform', 'MSE', None, dict(num_candidates=8)),
                                         ('asymmetric_uniform', 'cross_entropy', 'logits', dict(num_candidates=8))):
            mgr = QuantizationManager(qmethod=QMethods[method], init=RangeEstimators[init],
                                      qparams=dict(n_bits=4 if init == 'MSE' else 8), init_params=ip)
            if layout == 'per_embd':
                set_act_quant_axis_and_groups(mgr, axis=2, n_groups=None)
            for x in xs:
                mgr(x[:, 0, :3].contiguous() if layout == 'logits' else x)
            ref.append((mgr.range_estimator.current_xmin.clone(), mgr.range_estimator.current_xmax.clone(),
                        mgr.quantizer._delta.clone()))
        with pytest.raises(Exception):          # not sharded: an empty tensor stays an error, as in the reference
            QuantizationManager(qmethod=QMethods.asymmetric_uniform, init=RangeEstimators.current_minmax,
                                qparams=dict(n_bits=8))(torch.zeros(0, 4))
    finally:
        _hip.set_backend(prev)
    for a, b, r in zip(r0, r1, ref):
        for k in range(3):
            assert torch.equal(a[k], b[k]) and torch.equal(a[k], r[k])


def _one_layer_bert():
    from transformers import BertConfig, BertForSequenceClassification
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from tests.harness_bert import QBertForSequenceClassification
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
              weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.current_minmax)
    torch.manual_seed(1000)
    cfg = BertConfig(num_labels=2, hidden_size=64, num_attention_heads=4, intermediate_size=128, vocab_size=1000,
                     num_hidden_layers=1, max_position_embeddings=64)
    model = QBertForSequenceClassification(BertForSequenceClassification(cfg).eval(), **qp)
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(10, 1000, (1, 16), generator=g)                     # ONE calibration sample
    return model.eval(), ids


def _bert_ranges(model):
    from tests.harness_bert import quantizer_census
    act, _ = quantizer_census(model)
    return torch.stack([torch.stack([m.range_estimator.current_xmin.reshape(()), m.range_estimator.current_xmax.reshape(())])
                        for _, m in act if m.quantizer.is_initialized])


def _worker_bert_one_sample(rank, port, outdir):
    _setup(rank, port)
    from utils.utils import pass_data_for_range_estimation
    model, ids = _one_layer_bert()
    with torch.no_grad():
        pass_data_for_range_estimation([(ids,)], model, act_quant=True, weight_quant=True, max_num_batches=1)
    torch.save(_bert_ranges(model), os.path.join(outdir, f'bert1_{rank}.pt'))
    dist.destroy_process_group()


def test_whole_model_calibration_with_one_sample_over_two_ranks(tmp_path):
    """The README recipe's calibration (ONE sample, utils.pass_data_for_range_estimation) sharded over 2 ranks: rank 1
    runs the whole model on an empty batch, and both ranks end with the single-process activation ranges."""
    port = _free_port()
    mp.spawn(_worker_bert_one_sample, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    from quantization import _hip, distributed as tq_dist
    from tests._oracle_backend import OracleBackend
    from utils.utils import pass_data_for_range_estimation
    prev = _hip.set_backend(OracleBackend())
    try:
        tq_dist.disable()
        model, ids = _one_layer_bert()
        with torch.no_grad():
            pass_data_for_range_estimation([(ids,)], model, act_quant=True, weight_quant=True, max_num_batches=1)
        ref = _bert_ranges(model)
    finally:
        _hip.set_backend(prev)
    r0 = torch.load(os.path.join(tmp_path, 'bert1_0.pt'), weights_only=False)
    r1 = torch.load(os.path.join(tmp_path, 'bert1_1.pt'), weights_only=False)
    assert r0.shape == ref.shape and r0.shape[0] > 10
    assert torch.equal(r0, r1) and torch.equal(r0, ref)


# ---- data-parallel QAT step: bucketed gradient all-reduce (BASELINE configs[4]) ---------------------
def _qat_problem():
    """Toy quantized model, calibrated on the GLOBAL batches with the exchange off (identical on every rank), ranges
    learnable (reference utils/qat_utils.py:26-28: learn_ranges after range estimation)."""
    from quantization import distributed as tq_dist
    from utils.utils import pass_data_for_range_estimation
    torch.manual_seed(5)
    model = _quant_toy(_Toy())
    batches = _batches(n=4, B=8, seed=3)
    with tq_dist.suspended():
        pass_data_for_range_estimation([(b,) for b in batches[:2]], model, act_quant=True, weight_quant=True,
                                       max_num_batches=2)
    model.learn_ranges()
    model.set_quant_state(True, True)
    model.train()
    g = torch.Generator().manual_seed(9)
    targets = [torch.randn(8, 6, 24, generator=g) for _ in batches]
    return model, batches, targets


def _qat_run(model, batches, targets, shard, opt_name, bucket_bytes, steps=6, first=0, accumulate_over=None):
    """accumulate_over=N: ONE process walks the N shards of every batch, accumulating their gradients in the buckets and
    averaging -- arithmetically the data-parallel step (same per-shard GEMM shapes, g0 + g1 is commutative)."""
    from quantization.data_parallel import GradientBuckets, train_step
    params = [p for p in model.parameters() if p.requires_grad]
    opt = (torch.optim.SGD(params, lr=0.05, momentum=0.9) if opt_name == 'sgd'
           else torch.optim.Adam(params, lr=1e-2))
    gb = GradientBuckets(params, bucket_bytes=bucket_bytes)
    losses, grads = [], None
    for i in range(first, first + steps):
        xb, tb = batches[i % len(batches)], targets[i % len(batches)]
        if accumulate_over:
            gb.zero_()
            per = xb.shape[0] // accumulate_over
            ls = []
            for r in range(accumulate_over):
                loss = torch.nn.functional.mse_loss(model(xb[r * per:(r + 1) * per]), tb[r * per:(r + 1) * per])
                loss.backward()
                ls.append(float(loss.detach()))
            for f in gb._flats:
                f.mul_(1.0 / accumulate_over)
            opt.step()
            losses.append(ls)
        else:
            losses.append(float(train_step(model, torch.nn.functional.mse_loss, opt, gb, (shard(xb),), (shard(tb),))))
        if grads is None:
            grads = {k: v.grad.detach().clone() for k, v in model.named_parameters()}
    out = {k: v.detach().clone() for k, v in model.named_parameters()}
    return out, losses, gb.n_buckets, gb.launched, grads


_QAT_CASES = [('sgd', 1 << 30, 6, 0), ('sgd', 2048, 1, 1), ('adam', 2048, 6, 2)]


def _worker_qat(rank, port, outdir):
    tq_dist = _setup(rank, port)
    res = []
    for opt_name, bucket_bytes, steps, first in _QAT_CASES:
        model, batches, targets = _qat_problem()
        res.append(_qat_run(model, batches, targets, tq_dist.shard_batch, opt_name, bucket_bytes, steps, first))
    torch.save(res, os.path.join(outdir, f'qat_{rank}.pt'))
    dist.destroy_process_group()


def test_data_parallel_qat_equals_single_rank_on_the_concatenated_batch(tmp_path):
    """Two ranks, each on its half of every batch, gradients of weights AND learnable ranges summed in flat buckets and
    averaged.

    * BIT-EXACT over six optimizer steps (SGD-momentum with one bucket, Adam with many small buckets) against ONE process
      that accumulates the gradients of the same two half-batches and averages them -- the same arithmetic, so any
      difference would be a bug in bucketing / hook order / the exchange, not round-off;
    * against one process on the WHOLE batch (mean-reduced loss) the first-step gradients agree to the round-off a
      different GEMM shape causes in a discontinuous network: the 8-row and the 4-row products differ in the last bit,
      a few activations land on the other side of a rounding boundary and their idx-weighted terms move the range
      gradients by up to a few 1e-3 relative (weights 1e-4); trajectories are not compared beyond that point."""
    port = _free_port()
    mp.spawn(_worker_qat, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    from quantization import _hip, distributed as tq_dist
    from tests._oracle_backend import OracleBackend
    prev = _hip.set_backend(OracleBackend())
    try:
        tq_dist.disable()
        got = [torch.load(os.path.join(tmp_path, f'qat_{r}.pt'), weights_only=False) for r in range(WORLD)]
        for ci, (opt_name, bucket_bytes, steps, first) in enumerate(_QAT_CASES):
            model, batches, targets = _qat_problem()
            start = {k: v.detach().clone() for k, v in model.named_parameters()}
            acc, acc_losses, n_buckets, launched, acc_g = _qat_run(model, batches, targets, None, opt_name, bucket_bytes,
                                                                   steps, first, accumulate_over=WORLD)
            assert launched == 0                                  # single process: storage only, no collective
            ranges = [k for k in acc if k.endswith('_delta') or k.endswith('_zero_float')]
            assert len(ranges) >= 6, 'learnable ranges must be among the trained parameters'
            p0, l0, nb0, launched0, g0 = got[0][ci]
            p1, l1, nb1, launched1, g1 = got[1][ci]
            assert nb0 == nb1 == n_buckets and launched0 == launched1 == steps * n_buckets
            assert (n_buckets == 1) if bucket_bytes > 2048 else (n_buckets > 2)
            for k in acc:
                assert torch.equal(p0[k], p1[k]) and torch.equal(g0[k], g1[k]), (ci, k)   # replicas stay bit-identical
                assert torch.equal(g0[k], acc_g[k]), (ci, k, 'first-step gradient')
                assert torch.equal(p0[k], acc[k]), (ci, k, f'parameters after {steps} steps')
            assert [[a, b] for a, b in zip(l0, l1)] == acc_losses
            moved = sum(int(not torch.equal(start[k], acc[k])) for k in ranges)
            assert moved >= len(ranges) // 2, 'the range parameters must actually train'
            # semantic check: == the gradient of the mean loss over the concatenated batch
            model, batches, targets = _qat_problem()
            _, full_losses, _, _, full_g = _qat_run(model, batches, targets, lambda x: x, opt_name, bucket_bytes, 1, first)
            assert abs((l0[0] + l1[0]) / 2 - full_losses[0]) <= 1e-5 * abs(full_losses[0]) + 1e-7
            for k in acc:
                scale = float(full_g[k].abs().max()) + 1e-12
                tol = 1e-2 if k in ranges else 5e-4
                assert float((g0[k] - full_g[k]).abs().max()) <= tol * scale + 1e-9, (ci, k)
    finally:
        _hip.set_backend(prev)


def test_gradient_buckets_refuse_detached_gradients():
    from quantization.data_parallel import GradientBuckets
    lin = torch.nn.Linear(4, 3)
    gb = GradientBuckets(lin.parameters(), bucket_bytes=16)
    assert gb.n_buckets == 2 and sorted(gb.bucket_sizes()) == [12, 48]
    lin(torch.ones(2, 4)).sum().backward()
    assert torch.equal(lin.bias.grad, torch.full((3,), 2.0)) and gb.launched == 0
    gb.finish()
    gb.zero_()
    assert float(lin.weight.grad.abs().sum()) == 0.0
    torch.optim.SGD(lin.parameters(), lr=0.1).zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError, match='detached from its bucket'):
        gb.zero_()
