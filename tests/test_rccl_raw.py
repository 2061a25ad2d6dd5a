"""The raw-RCCL exchange behind the C ABI (tq_comm_* / tq_calibrate_minmax_rccl, quantization/rccl.py): sharded
calibration with NO c10d collective in the data path.  A GPU box of the pool has one device and RCCL refuses two ranks
per device, so the communicator has one rank here; world-size-2 semantics of the same hooks are covered on CPU with gloo
(tests/test_dist_gloo.py).  The workers initialise torch.distributed with the `gloo` backend (or not at all): there is
no NCCL watchdog thread in these processes, which is the point -- the calibrating forward, collectives included, is
captured as a hipGraph repeatedly."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_comm_entry_points_bind_librccl_without_a_gpu():
    """dlopen + dlsym only (no device call): the library binds the librccl torch ships and reports its version."""
    sys.path.insert(0, os.path.join(ROOT, 'transformer-quantization_amd'))
    from quantization import _hip, rccl
    lib = _hip.load_library()
    assert lib.tq_comm_unique_id_bytes() == 128
    path = rccl._librccl_path()
    assert path and os.path.exists(path)
    assert lib.tq_comm_load(path.encode()) == 0
    assert lib.tq_comm_version() > 20000
    assert lib.tq_comm_allreduce(None, None, 4, 0, 0, None) == -1 and b'NULL' in lib.tq_last_error()
    assert lib.tq_comm_destroy(None) == 0


PRELUDE = r'''
import os, sys
sys.path.insert(0, os.path.join(ROOT, 'transformer-quantization_amd')); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
'''

GRAPH_WORKER = PRELUDE + r"""
dist.init_process_group('gloo')                    # rendezvous store only: no c10d collective below
from quantization import distributed as tq_dist, options
from quantization.graphs import GraphedForward
from tests.test_calibration_graph import _model, _batches
batches = _batches(4)
tq_dist.enable(force=True, raw=True)
assert tq_dist.raw_comm() is not None and tq_dist.raw_comm().version > 20000
with torch.no_grad():
    ref = _model(2)
    for b in batches:
        ref_out = ref(tq_dist.shard_batch(b))
    ref_sd = {k: v.clone() for k, v in ref.state_dict().items()}
    st = tq_dist.stats()
    assert st['raw_rccl_calls'] >= 4 * 27 and st['raw_rccl_calls'] == st['minmax_calls'], st   # every exchange was raw
    options.INPLACE_CALIBRATION_STATE = True
    for rep in range(int(os.environ.get('TQ_TEST_CAPTURES', '5'))):
        m = _model(2)
        m(batches[0])                                  # first batch eager: allocates every state buffer
        n0 = tq_dist.stats()['minmax_calls']
        g = GraphedForward(m, batches[1])              # captures the ncclAllReduce launches with the kernels
        n1 = tq_dist.stats()['minmax_calls']
        for b in batches[1:]:
            out = g(b)
        assert tq_dist.stats()['minmax_calls'] == n1 and n1 > n0      # replay issues no python-side collective call
        sd = m.state_dict()
        assert sd.keys() == ref_sd.keys()
        for k in sd:
            assert torch.equal(sd[k], ref_sd[k]), (rep, k)
        assert torch.equal(out, ref_out), rep
        del g, m
tq_dist.disable()
dist.destroy_process_group()
print('RAW_RCCL_GRAPH_CALIBRATION_OK')
"""

NO_PG_WORKER = PRELUDE + r"""
# no torch.distributed process group at all: the ncclUniqueId travels through a TCPStore on MASTER_ADDR / MASTER_PORT+1
from quantization.rccl import RawRcclComm, MAX, SUM
comm = RawRcclComm()
assert (comm.rank, comm.world) == (0, 1) and comm.self_test()
v = torch.tensor([-1.0, 2.5, 3.0], device='cuda')
assert comm.allreduce_(v.clone(), MAX).tolist() == v.tolist()
d = torch.arange(101, dtype=torch.float64, device='cuda')
assert torch.equal(comm.allreduce_(d.clone(), SUM), d)
b = torch.arange(7, dtype=torch.int32, device='cuda')
assert torch.equal(comm.broadcast_(b.clone(), 0), b)
import ctypes as C
r, w = C.c_int(-1), C.c_int(-1)
assert comm.lib.tq_comm_rank_world(comm.handle, C.byref(r), C.byref(w)) == 0 and (r.value, w.value) == (0, 1)
comm.close()
print('RAW_RCCL_NO_PG_OK')
"""


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(script, extra_env=None, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **(extra_env or {}))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), str(script)]
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.gpu
def test_sharded_calibration_replays_as_hipgraph_with_raw_rccl(tmp_path):
    """A sharded calibrating forward -- statistics kernel -> ncclAllReduce(MAX) in place -> update + quantize at every
    site, each site ONE C call -- captured as a hipGraph (collectives included) and replayed per batch == eager sharded
    calibration, bit for bit (state_dict and outputs); five captures in one process."""
    script = tmp_path / 'graph_worker.py'
    script.write_text('ROOT = %r\n' % ROOT + GRAPH_WORKER)
    r = _torchrun(script)
    assert r.returncode == 0 and 'RAW_RCCL_GRAPH_CALIBRATION_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_raw_comm_without_a_process_group(tmp_path):
    script = tmp_path / 'nopg_worker.py'
    script.write_text('ROOT = %r\n' % ROOT + NO_PG_WORKER)
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(port), TQ_RCCL_STORE_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, str(script)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'RAW_RCCL_NO_PG_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
