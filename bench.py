#!/usr/bin/env python3
"""bench.py -- fake-quant forward throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the fixed-range quantize->clip->dequantize op (W8A8 activation quantizer,
asymmetric 8-bit, per-tensor, range from the running-min/max estimator) over one batch of synthetic
BERT-base hidden states [B, S, 768] bf16 that is already resident in HBM.  The op is called through
the drop-in class API (QuantizedActivation -> QuantizationManager -> AsymmetricUniformQuantizer ->
ctypes -> libtq_hip.so), not through a private fast path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--seq S] [--sweep]

N > 1: one rank per GPU over RCCL.  Either the caller wraps the command in torch.distributed.run (RANK /
LOCAL_RANK / WORLD_SIZE in the environment), or -- when `--gpus N` is given without such an environment --
bench.py re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1`.  Each rank owns its own [B, S, 768] shard (weak scaling, no data-path collective
in the fixed-range forward).  The calibration phase before the timed region DOES exchange statistics: one
fused MAX all-reduce of [-min; max] per quantizer call; its throughput at N ranks is reported under
"calibration" (the north-star's 1/2/4/8-GPU calibration throughput).

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'transformer-quantization_amd')
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_ELEM = 4             # bf16 in + bf16 out: algorithmic bytes of K1
D_MODEL = 768


def make_hidden(B, S, device, seed, dtype=torch.bfloat16):
    """Synthetic BERT-base hidden state: unit normal, embedding dims 308 and 381
    scaled x20 on every token and x60 on the last token."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn(B, S, D_MODEL, generator=g, device=device, dtype=torch.float32)
    for d in (308, 381):
        x[..., d] *= 20.0
        x[:, -1, d] *= 3.0
    return x.to(dtype)


def _physical_cores():
    """Physical core count from /proc/cpuinfo ((physical id, core id) pairs); logical count / 2 as a fallback."""
    try:
        pairs, phys, core = set(), None, None
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('physical id'):
                    phys = line.split(':', 1)[1].strip()
                elif line.startswith('core id'):
                    core = line.split(':', 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(budget_s=14.0):
    """The reference's PyTorch-CPU quantizer path (oracle port, bit-identical to the reference per
    tests/test_oracle_golden.py), fp32, on this box's host cores, at TWO sizes of the same workload (BASELINE.md
    section 4): the cache-resident config shape [8, 128, 768] (where the CPU is at its best: no page faults, the op
    chain's temporaries stay in L2/L3) and a bounded [64, 512, 768] sample of the large tensor (every ATen op of the
    chain allocates a fresh 100 MB tensor: page-fault bound).  `value` is the BEST rate over both sizes and all thread
    counts -- the baseline most favourable to the CPU; everything else is reported beside it."""
    from oracle import tq_oracle as O
    threads = torch.get_num_threads()
    phys = _physical_cores()

    def hidden(b, s):
        g = torch.Generator().manual_seed(1000)
        x = torch.randn(b, s, D_MODEL, generator=g)
        x[..., 308] *= 20.0
        x[..., 381] *= 20.0
        x[:, -1, 308] *= 3.0
        x[:, -1, 381] *= 3.0
        return x

    def run(x, delta, zf, nthreads, budget, max_reps):
        torch.set_num_threads(nthreads)
        for _ in range(10 if x.numel() < (1 << 22) else 2):
            O.fake_quant(x, delta, zf, 8, False)
        times = []
        t_end = time.perf_counter() + budget
        while time.perf_counter() < t_end and len(times) < max_reps:
            t0 = time.perf_counter()
            O.fake_quant(x, delta, zf, 8, False)
            times.append(time.perf_counter() - t0)
        times.sort()
        return x.numel() / times[len(times) // 2] / 1e6, len(times)

    limit = max(threads, phys)
    counts = sorted({c for c in (1, 8, 16, 32, 64, phys, threads) if 1 <= c <= limit})
    points = {}
    for name, (b, s), share, max_reps in (('config_shape', (8, 128), 0.3, 400), ('large_sample', (64, 512), 0.7, 200)):
        x = hidden(b, s)
        delta, zf = O.asym_params_from_range(x.min(), x.max(), 8)
        per, reps_by = {}, {}
        for c in counts:
            v, reps = run(x, delta, zf, c, budget_s * share / len(counts), max_reps)
            per[str(c)] = round(v, 1)
            reps_by[str(c)] = reps
        best = max(per, key=lambda k: per[k])
        points[name] = {'shape': [b, s, D_MODEL], 'elems': x.numel(), 'M_elems_s_by_threads': per,
                        'best_threads': int(best), 'best_M_elems_s': per[best], 'median_of': reps_by[best]}
    torch.set_num_threads(threads)
    win = max(points, key=lambda k: points[k]['best_M_elems_s'])
    w = points[win]
    return {
        'value': w['best_M_elems_s'], 'unit': 'M elems/s', 'cores': w['best_threads'], 'kind': 'port',
        'sample': f'{w["shape"]} fp32 hidden states ({w["elems"]} elems, {win}), fixed-range asym 8-bit fake-quant '
                  f'(reference op chain, oracle port), median of {w["median_of"]} passes, torch {torch.__version__} CPU; '
                  f'best over thread counts {counts} and over the two sizes in `points`',
        'points': points,
        'physical_cores': phys, 'host_logical_cpus': os.cpu_count(), 'cpu_model': _cpu_model(),
    }


PMC_TRAFFIC_JSON = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')


def pmc_traffic(n_elems):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KiB -> bytes).
    None when no PMC summary exists for this exact workload size."""
    path = PMC_TRAFFIC_JSON
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    if t.get('workload_elems') != n_elems:
        return None
    return t.get('traffic_bytes_per_launch')


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def timed_region(fn, steps, use_dist):
    """barrier + sync, K back-to-back steps bracketed by ONE HIP event pair on torch's current
    stream (the stream the kernels are launched on), sync + barrier.
    -> (wall seconds for the K steps, device milliseconds per step = event span / K).
    The launches are asynchronous and the host stays ahead (≈15 us of Python per 250 us kernel), so
    the span is K kernels plus K-1 launch boundaries of ≈1.5 us -- it agrees with rocprofv3's
    per-kernel average to < 1 %."""
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    start.record()
    for _ in range(steps):
        fn()
    end.record()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    wall = time.perf_counter() - t0
    return wall, start.elapsed_time(end) / steps


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` without a torchrun environment: spawn the N ranks ourselves."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault('OMP_NUM_THREADS', '4')
    raise SystemExit(subprocess.call(cmd, env=env))


def dry_run(args, rank, world):
    """TQ_BENCH_DRY_RUN=1: launcher / rendezvous / reporting control flow WITHOUT a GPU and without any kernel
    (CPU test of `--gpus N`, tests/test_dist_gloo.py).  The line it prints is labelled and carries no measurement."""
    if world > 1 or 'RANK' in os.environ:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')
        dist.barrier()
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t[0]) == world
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({'metric': 'DRY RUN (no kernels launched, not a measurement)', 'value': None, 'dry_run': True,
                          'n_gpus': world, 'rccl_world_size': world, 'steps': args.steps, 'warmup': args.warmup}),
              flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--seq', type=int, default=512)
    ap.add_argument('--sweep', action='store_true', help='also time the SURVEY.md 8d shape sweep')
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--mailbox', action='store_true', help='also time calibration with the P2P mailbox exchange')
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ:
        relaunch_under_torchrun(args.gpus)

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if os.environ.get('TQ_BENCH_DRY_RUN') == '1':
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU (there is no CPU fallback in the product path)')
    # TQ_BENCH_SAME_DEVICE=1 + TQ_BENCH_BACKEND=gloo: control-flow test of the N>1 path on a
    # 1-GPU box (every rank on cuda:0, host-staged collectives).  Never used for reported numbers.
    if os.environ.get('TQ_BENCH_SAME_DEVICE') == '1':
        local_rank = 0
    backend = os.environ.get('TQ_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    # TQ_BENCH_FORCE_DIST=1: initialise the process group (RCCL) even for a 1-rank torchrun launch, so the
    # collective path (barriers, fused MAX all-reduce in calibration) can be exercised on a 1-GPU box.
    use_dist = world > 1 or (os.environ.get('TQ_BENCH_FORCE_DIST') == '1' and 'RANK' in os.environ)
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)

    from quantization import _hip, distributed as tq_dist
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from quantization.base_quantized_classes import QuantizedActivation
    assert _hip.backend().name == 'hip'
    if use_dist:
        # device tensors are exchanged on the raw RCCL communicator inside libtq_hip.so when the backend is `nccl`
        # (quantization/rccl.py; self-tested at creation); torch.distributed is the fallback
        try:
            tq_dist.enable(force=(world == 1))
        except Exception as e:       # noqa: BLE001
            print(f'[bench] raw RCCL exchange unavailable ({e!r}): statistics go through torch.distributed', file=sys.stderr)
            tq_dist.enable(force=(world == 1), raw=False)
    transport = 'none'
    if use_dist:
        transport = 'raw RCCL (tq_calibrate_minmax_rccl)' if tq_dist.raw_comm() is not None else f'torch.distributed ({backend})'

    B, S = args.batch, args.seq
    x = make_hidden(B, S, device, seed=1000 + rank)
    n_elems = x.numel()

    qa = QuantizedActivation(act_method=QMethods.asymmetric_uniform, n_bits_act=8,
                             act_range_method=RangeEstimators.running_minmax).to(device)
    qa.quantized_acts()
    qa.eval()

    # ---- calibration (untimed for `value`; reported separately): estimate + quantize ----------
    calib_batches = [x, make_hidden(B, S, device, seed=2000 + rank)]
    for xb in calib_batches:
        qa(xb)
    cal_wall, cal_ms = timed_region(lambda: qa(x), max(4, min(args.steps, 20)), use_dist)
    cal_steps = max(4, min(args.steps, 20))
    # --mailbox / TQ_BENCH_MAILBOX=1: the same step with the statistics exchanged through the P2P mailbox kernel instead
    # of ncclAllReduce (reported next to the RCCL figure, never instead of it; skipped if the set-up or the self-test
    # against RCCL fails).  Opt-in: the path has been validated with two processes on one device and with a 1-rank RCCL
    # group only (no multi-GPU box was available to the build), and the headline run must not depend on it.
    mail_wall = None
    if use_dist and backend == 'nccl' and (args.mailbox or os.environ.get('TQ_BENCH_MAILBOX', '0') == '1'):
        try:
            tq_dist.enable(force=(world == 1), mailbox=True, raw=tq_dist.raw_comm() is not None)
            if tq_dist.mailbox_active():
                qa(x)
                mail_wall, _ = timed_region(lambda: qa(x), cal_steps, use_dist)
        except Exception as e:       # noqa: BLE001
            print(f'[bench] mailbox leg skipped: {e!r}', file=sys.stderr)
        finally:
            tq_dist.enable(force=(world == 1), mailbox=False, raw=tq_dist.raw_comm() is not None)
    qa.activation_quantizer.fix_ranges()
    del calib_batches

    # ---- the hot path: fixed-range fake-quant forward -------------------------------------------
    with torch.no_grad():
        # untimed: let the power management settle (the first ~100 ms after an idle period run at a
        # lower clock: 280-340 us per launch vs 255-265 us steady state for this kernel)
        t_settle = time.perf_counter() + 0.4
        while time.perf_counter() < t_settle:
            for _ in range(20):
                qa(x)
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            qa(x)
        wall, ev_ms = timed_region(lambda: qa(x), args.steps, use_dist)

    if use_dist:
        tmax = torch.tensor([wall, cal_wall, mail_wall if mail_wall is not None else -1.0], device=device,
                            dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall, cal_wall = float(tmax[0]), float(tmax[1])
        mail_wall = float(tmax[2]) if mail_wall is not None else None

    total_elems = n_elems * world * args.steps
    value = total_elems / wall / 1e6
    kernel_s = ev_ms / 1e3
    achieved = n_elems * BYTES_PER_ELEM / kernel_s / 1e9

    out = {
        'metric': 'M elems/sec fake-quant fwd (BERT-base act tensor); achieved HBM GB/s vs peak',
        'value': round(value, 1),
        'unit': 'M elems/s',
        'n_gpus': world,
        'rccl_world_size': dist.get_world_size() if use_dist else 1,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': round(wall / args.steps * 1e3, 4),
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',   # arithmetic type of the path: fp32 register math on bf16 storage
        'data': 'synthetic',
        'config': {
            'workload': f'BERT-base W8A8 per-tensor asym 8-bit activation fake-quant, '
                        f'running-minmax calibrated, fixed-range fwd, hidden [{B},{S},768] bf16 '
                        f'per GPU (BASELINE configs[1])',
            'storage_dtype': 'bf16', 'elems_per_gpu_per_step': n_elems,
            'parallelism': f'dp{world} (independent shards, no data-path collective)',
        },
        'roofline': {
            'bound': 'hbm',
            'achieved': round(achieved, 1),
            'peak': HBM_PEAK_GBS,
            'unit': 'GB/s',
            'frac': round(achieved / HBM_PEAK_GBS, 4),
            'traffic': pmc_traffic(n_elems),
            'traffic_source': 'committed profile, NOT a counter of this run: bytes per launch from '
                              'profiles/pmc_traffic.json (rocprofv3 --pmc passes of this same command, gfx950 '
                              'FETCH_SIZE x2 correction); null if not collected for this workload size',
            'kernel': 'tq::fq_tensor<bf16>',
            'kernel_ms': round(ev_ms, 4),
            'algorithmic_bytes_per_launch': n_elems * BYTES_PER_ELEM,
        },
        'calibration': {
            'what': 'estimate (tq_minmax -> range_update -> set_range) + quantize per step; '
                    + ('one fused MAX all-reduce of [-min;max] per step over RCCL' if world > 1
                       else 'single GPU, no collective'),
            'transport': transport,
            'value': round(n_elems * world * cal_steps / cal_wall / 1e6, 1),
            'unit': 'M elems/s',
            'ms_per_step': round(cal_wall / cal_steps * 1e3, 4),
        },
    }
    if mail_wall is not None:
        out['calibration']['p2p_mailbox'] = {
            'what': 'same step, [-min;max] exchanged by the P2P mailbox kernel (tq_mailbox_allreduce_max) instead of RCCL',
            'value': round(n_elems * world * cal_steps / mail_wall / 1e6, 1), 'unit': 'M elems/s',
            'ms_per_step': round(mail_wall / cal_steps * 1e3, 4)}

    if args.sweep and rank == 0:
        sweep = []
        for (b, s) in [(8, 128), (64, 128), (256, 128), (256, 512), (1024, 512)]:
            xs = make_hidden(b, s, device, seed=7)
            with torch.no_grad():
                for _ in range(5):
                    qa(xs)
                _, ms = timed_region(lambda: qa(xs), 30, False)
            sweep.append({'shape': [b, s, 768], 'kernel_ms': round(ms, 4),
                          'M_elems_s': round(xs.numel() / ms / 1e3, 1),
                          'GBps': round(xs.numel() * BYTES_PER_ELEM / ms / 1e6, 1)})
        out['sweep'] = sweep

    if rank == 0 and world == 1 and not args.no_cpu:
        out['cpu_baseline'] = cpu_baseline()
    elif rank == 0:
        out['cpu_baseline'] = None

    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
