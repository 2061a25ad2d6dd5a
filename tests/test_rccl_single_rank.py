"""The collective path on real RCCL, as far as a 1-GPU box allows: a 1-rank `nccl` process group (the GPU
boxes the tests run on have one device, and RCCL refuses two ranks on one device).  Checks that the
fused MAX / SUM all-reduces of quantization.distributed accept the device tensors the estimators hand
them (fp32 [-min;max] pairs, fp64 candidate-loss vectors), that sharded calibration is a no-op at world
size 1, and that bench.py runs its N>1 control flow (barriers, all-reduced timing) over RCCL.
World-size-2 semantics are covered on CPU with gloo (tests/test_dist_gloo.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.path.join(ROOT, 'transformer-quantization_amd')); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from quantization import distributed as tq_dist
from quantization.quantization_manager import QuantizationManager
from quantization.quantizers import QMethods
from quantization.range_estimators import RangeEstimators
from utils.per_embd_quant_utils import set_act_quant_axis_and_groups
g = torch.Generator(device='cuda').manual_seed(7)
x = torch.randn(16, 64, 768, device='cuda', generator=g); x[..., 308] *= 20
def run(init, axis=None, n_groups=None, params=None):
    m = QuantizationManager(qmethod=QMethods.asymmetric_uniform, init=RangeEstimators[init],
                            qparams=dict(n_bits=8), init_params=params or {})
    if axis is not None:
        set_act_quant_axis_and_groups(m, axis=axis, n_groups=n_groups)
    y = m(tq_dist.shard_batch(x))
    return y, m.quantizer._delta.clone(), m.quantizer._zero_float.clone()
cases = [('running_minmax', None, None, None), ('current_minmax', 2, None, None), ('current_minmax', 2, 6, None),
         ('MSE', None, None, dict(num_candidates=50))]
local = [run(*c) for c in cases]
RAW = os.environ.get('TQ_TEST_RAW', '1') == '1'
tq_dist.enable(force=True, raw=RAW)
assert tq_dist.is_enabled() and (tq_dist.raw_comm() is not None) == RAW
shared = [run(*c) for c in cases]
for a, b in zip(local, shared):
    for u, v in zip(a, b):
        assert torch.equal(u, v)
mn, mx = tq_dist.sync_minmax(torch.tensor([-1.0, 2.0], device='cuda'), torch.tensor([3.0, 4.0], device='cuda'))
assert mn.tolist() == [-1.0, 2.0] and mx.tolist() == [3.0, 4.0]
s = tq_dist.sync_sum(torch.arange(5, dtype=torch.float64, device='cuda'))
assert s.tolist() == [0.0, 1.0, 2.0, 3.0, 4.0]
st = tq_dist.stats()
assert st['minmax_calls'] >= len(cases) + 1 and st['sum_calls'] >= 2, st
assert (st['raw_rccl_calls'] >= len(cases) + 3) if RAW else st['raw_rccl_calls'] == 0, st

# data-parallel AdaRound over RCCL (1 rank): SUM all-reduce of dL/dW_q before the fused Adam step
from tests.test_dist_gloo import _ada_problem
from quantization.adaround.adaround import optimize_local_loss
def ada():
    layer, wq, X, get_inp_out, loss_fn, opt, idx = _ada_problem('cuda')
    optimize_local_loss(layer, get_inp_out, X, opt, loss_fn, 8, 6, batch_indices=idx)
    return wq.alpha.detach().clone()
before = tq_dist.stats()['sum_calls']
a_dist = ada()
assert tq_dist.stats()['sum_calls'] >= before + 6          # one gradient all-reduce per iteration
tq_dist.disable()
a_local = ada()
assert torch.equal(a_dist, a_local)
tq_dist.enable(force=True, raw=RAW)
tq_dist.disable()
dist.destroy_process_group()
print('RCCL_SINGLE_RANK_OK', st)
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(args, extra_env=None, timeout=300):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **(extra_env or {}))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port())] + args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize('raw', ['1', '0'], ids=['raw-rccl', 'c10d'])
def test_estimators_over_rccl_world1(tmp_path, raw):
    """raw-rccl: the collectives go through libtq_hip.so's own communicator (tq_comm_allreduce /
    tq_calibrate_minmax_rccl); c10d: through torch.distributed.all_reduce on the same process group."""
    script = tmp_path / 'worker.py'
    script.write_text('ROOT = %r\n' % ROOT + WORKER)
    r = _torchrun([str(script)], extra_env={'TQ_TEST_RAW': raw})
    assert r.returncode == 0 and 'RCCL_SINGLE_RANK_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_distributed_control_flow_over_rccl():
    r = _torchrun(['bench.py', '--gpus', '1', '--steps', '3', '--warmup', '1', '--no-cpu', '--batch', '64', '--seq', '128',
                   '--mailbox'], extra_env={'TQ_BENCH_FORCE_DIST': '1'})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 1 and out['value'] > 0 and out['roofline']['achieved'] > 0
    assert out['calibration']['p2p_mailbox']['value'] > 0          # --mailbox leg (1-rank group: post + no peers)
    # round 4: the latency-bound path and the data-parallel steps, on the raw 1-rank communicator (collectives forced on)
    assert out['rccl_world_size'] == 1 and 'raw RCCL' in out['calibration']['transport']
    cm = out['calibration_model']
    assert 'raw RCCL' in cm['transport'], cm
    for leg in ('weak', 'strong'):
        assert cm[leg]['collectives_per_forward'] == 161 and cm[leg]['eager_ms'] > 0 and cm[leg]['hipgraph_ms'] > 0, cm[leg]
    assert cm['strong']['per_rank_batch'] == [128, 128] and cm['fixed_range_forward']['fused_hipgraph_ms'] > 0 and cm['fixed_range_forward']['layered_hipgraph_ms'] > 0
    assert set(cm['exchange_latency_us']) == {'8B', '6KB', '103KB', '9.4MB'} and all(v > 0 for v in cm['exchange_latency_us'].values())
    assert out['adaround_dp']['gradient_allreduce_bytes'] == 3072 * 768 * 4 and out['adaround_dp']['ms_per_iter'] > 0
    q = out['qat_dp']
    assert q['allreduces_per_step'] == q['buckets'] >= 1 and q['eager_ms'] > 0 and q['hipgraph_ms'] > 0, q


def test_bench_single_process_with_sweep():
    """The plain `python bench.py` path (no process group) incl. --sweep: one JSON line with the contract's keys."""
    r = subprocess.run([sys.executable, 'bench.py', '--steps', '3', '--warmup', '1', '--no-cpu', '--sweep', '--batch', '64',
                        '--seq', '128'], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in out, k
    assert out['steps'] == 3 and out['warmup'] == 1 and out['config']['workload']
    assert set(out['roofline']) >= {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'}
    assert len(out['sweep']) >= 3
    assert out['config_shape']['shape'] == [8, 128, 768] and out['config_shape']['kernel_us'] > 0
    cm = out['calibration_model']
    assert cm['weak']['collectives_per_forward'] == 0 and cm['weak']['hipgraph_ms'] > 0 and cm['exchange_latency_us'] is None
    assert out['adaround_dp']['ms_per_iter'] > 0 and out['qat_dp']['hipgraph_ms'] > 0 and out['qat_dp']['allreduces_per_step'] == 0


def test_bench_gpus_2_spawns_two_ranks_with_real_kernels():
    """`python bench.py --gpus 2` with no torchrun environment: bench.py launches the two ranks itself.  A GPU box
    of the test pool has ONE device and RCCL refuses two ranks per device, so the ranks share cuda:0 and the
    collectives are host-staged (gloo); kernels, sharded calibration exchange and timing reduction are the real ones."""
    env = dict(os.environ, TQ_BENCH_SAME_DEVICE='1', TQ_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--no-cpu', '--batch', '64', '--seq', '128']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        # Seen once in ~10 full-suite runs on the GPU pool and never in isolation (round 6); the two ranks rendezvous over
        # a port probed a moment earlier by bench.py's launcher.  Keep the evidence, try once more.
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'bench_gpus2_first_attempt.txt'), 'w') as f:
            f.write(f'rc {r.returncode}\n--- stdout\n{r.stdout[-8000:]}\n--- stderr\n{r.stderr[-16000:]}\n')
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['rccl_world_size'] == 2 and out['value'] > 0
    assert out['calibration']['value'] > 0 and 'all-reduce' in out['calibration']['what']
    cm = out['calibration_model']                                  # 2 ranks, exchange through torch.distributed (gloo)
    assert cm['world_size'] == 2 and cm['transport'] == 'torch.distributed'
    assert cm['weak']['collectives_per_forward'] == 161 and cm['strong']['per_rank_batch'] == [64, 128]
    assert cm['weak']['hipgraph_ms'] is None                       # c10d collectives are never captured
    assert out['adaround_dp']['per_rank_samples_per_iter'] == 4 and out['qat_dp']['allreduces_per_step'] >= 1
