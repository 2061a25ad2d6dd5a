#!/usr/bin/env python3
"""BERT-base calibrating forward (B=8, T=128; 161 activation + 102 weight quantizer calls): single-GPU fused path vs the
sharded-calibration path over RCCL.  Run under torch.distributed.run on the GPU box (one rank per GPU; a 1-GPU box gives
a 1-rank RCCL group with the collectives forced on, which exercises the same launches + one ncclAllReduce per call):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 scripts/sharded_calib_bench.py

Prints one JSON document (commit it under profiles/rNN/sharded_calibration.json)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'transformer-quantization_amd'), ROOT):
    sys.path.insert(0, p)

import torch
import torch.distributed as dist

rank = int(os.environ.get('RANK', 0))
world = int(os.environ.get('WORLD_SIZE', 1))
local = int(os.environ.get('LOCAL_RANK', 0))
torch.cuda.set_device(local)
dev = torch.device('cuda', local)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
dist.init_process_group('nccl', device_id=dev, rank=rank, world_size=world)

from quantization import distributed as tq_dist, quantization_manager as qm
from tests.test_bert_e2e import _build, _fixture


def wall(fn, n=20, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


z = _fixture()
model, _ = _build(dev)
ids_all = torch.from_numpy(z['input_ids']).to(dev)          # global batch 8 x 128
out = {'world_size': world, 'global_batch': list(ids_all.shape)}
with torch.no_grad():
    model.set_quant_state(True, True)
    model.estimate_ranges()
    out['single_gpu_fused_ms'] = wall(lambda: model(ids_all))
    from quantization import options
    from quantization.graphs import GraphedForward
    options.INPLACE_CALIBRATION_STATE = True
    model(ids_all)
    g1 = GraphedForward(model, ids_all)
    out['single_gpu_hipgraph_ms'] = wall(lambda: g1(ids_all))
    del g1
    options.INPLACE_CALIBRATION_STATE = False
    # ---- exchange through torch.distributed (c10d -> RCCL): the round-2 path, kept as the comparison -----------------
    tq_dist.enable(force=(world == 1), raw=False)
    ids = tq_dist.shard_batch(ids_all)
    out['local_batch'] = list(ids.shape)
    before = tq_dist.stats()
    out['sharded_c10d_eager_ms'] = wall(lambda: model(ids))
    after = tq_dist.stats()
    out['collectives_per_forward'] = (after['minmax_calls'] - before['minmax_calls']) / 23
    out['bytes_per_forward'] = (after['bytes'] - before['bytes']) / 23
    qm.FUSED_CALIBRATION = False                                # round-1 path: layered estimator + sync_minmax
    out['sharded_c10d_layered_ms'] = wall(lambda: model(ids))
    qm.FUSED_CALIBRATION = True
    tq_dist.disable()
    # ---- raw RCCL behind the C ABI (tq_calibrate_minmax_rccl: statistics -> ncclAllReduce -> update + quantize, one
    # C call per site, no c10d) -----------------------------------------------------------------------------------------
    tq_dist.enable(force=(world == 1), raw=True)
    out['raw_rccl_active'] = tq_dist.raw_comm() is not None
    out['rccl_version'] = tq_dist.raw_comm().version if tq_dist.raw_comm() else None
    before = tq_dist.stats()
    out['sharded_raw_rccl_eager_ms'] = wall(lambda: model(ids))
    after = tq_dist.stats()
    out['raw_rccl_calls_per_forward'] = (after['raw_rccl_calls'] - before['raw_rccl_calls']) / 23
    out['ratio_raw_eager_vs_single_eager'] = out['sharded_raw_rccl_eager_ms'] / out['single_gpu_fused_ms']
    # the same sharded forward captured as ONE hipGraph, the 161 ncclAllReduce launches included
    try:
        options.INPLACE_CALIBRATION_STATE = True
        model(ids)
        g2 = GraphedForward(model, ids)
        out['sharded_raw_rccl_hipgraph_ms'] = wall(lambda: g2(ids))
        out['ratio_sharded_graph_vs_single_graph'] = out['sharded_raw_rccl_hipgraph_ms'] / out['single_gpu_hipgraph_ms']
        out['ratio_sharded_graph_vs_single_eager'] = out['sharded_raw_rccl_hipgraph_ms'] / out['single_gpu_fused_ms']
        del g2
    except Exception as e:                                         # noqa: BLE001
        out['sharded_hipgraph_error'] = repr(e)[:500]
    options.INPLACE_CALIBRATION_STATE = False
    tq_dist.disable()
    # the same exchange through the P2P mailbox kernel instead of ncclAllReduce (one small kernel, no c10d host work)
    try:
        tq_dist.enable(force=(world == 1), mailbox=True, raw=False)
        out['mailbox_active'] = tq_dist.mailbox_active()
        model(ids)
        b0 = tq_dist.stats()['mailbox_calls']
        out['sharded_mailbox_eager_ms'] = wall(lambda: model(ids))
        out['mailbox_calls_per_forward'] = (tq_dist.stats()['mailbox_calls'] - b0) / 23
        options.INPLACE_CALIBRATION_STATE = True
        model(ids)
        g3 = GraphedForward(model, ids)
        out['sharded_mailbox_hipgraph_ms'] = wall(lambda: g3(ids))
        del g3
    except Exception as e:                                         # noqa: BLE001
        out['sharded_mailbox_error'] = repr(e)[:500]
    options.INPLACE_CALIBRATION_STATE = False
    tq_dist.disable()
    out['ratio_c10d_eager_vs_single_eager'] = out['sharded_c10d_eager_ms'] / out['single_gpu_fused_ms']
    # ---- at which GLOBAL batch does sharding over 8 GPUs start to pay?  t_1(B): one GPU calibrates the whole batch;
    # t_8(B): a rank calibrates B / 8 samples through the sharded path (measured here on the 1-rank communicator: all
    # launches + the ncclAllReduce enqueue of every site; on 8 ranks each of the 161 collectives additionally waits for
    # the slowest peer, `allreduce_us_assumed` per site, added below as a model term, not a measurement).
    if os.environ.get('TQ_BREAK_EVEN', '1') == '1' and world == 1:
        be_rows = []
        g = torch.Generator(device=dev).manual_seed(11)
        tq_dist.enable(force=True, raw=True)
        for B in (8, 32, 128, 512):
            ids_B = torch.randint(1000, 30000, (B, 128), device=dev, generator=g)
            ids_loc = ids_B[:max(B // 8, 1)]
            tq_dist_was = tq_dist.suspended()
            with tq_dist_was:
                t1 = wall(lambda: model(ids_B), n=8, w=2)
            t8 = wall(lambda: model(ids_loc), n=8, w=2)
            be_rows.append({'global_batch': B, 'single_gpu_eager_ms': t1, 'per_rank_batch': int(ids_loc.shape[0]),
                            'sharded_rank_eager_ms': t8})
        tq_dist.disable()
        out['break_even'] = {'rows': be_rows, 'allreduce_us_assumed': 20.0, 'collectives_per_forward': 161,
                             'note': 't_8 = sharded_rank_eager_ms + 161 x allreduce_us_assumed / 1000'}
t = torch.tensor([out['sharded_raw_rccl_eager_ms']], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
out['sharded_raw_rccl_eager_ms_max_over_ranks'] = float(t[0])
if rank == 0:
    print(json.dumps(out, indent=1))
dist.destroy_process_group()
