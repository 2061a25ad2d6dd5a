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

Rank 0 prints ONE JSON line.
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


KERNEL_SOURCES = ('csrc/tq_fake_quant.hip', 'csrc/tq_device.h', 'csrc/tq_host.h')      # what tq::fq_tensor is built from


def kernel_source_hash():
    """sha256 over the sources of the dominant kernel, as recorded in profiles/pmc_traffic.json by
    scripts/summarize_profiles.py when the PMC passes were collected."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, 'transformer-quantization_amd', rel), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def pmc_traffic(n_elems):
    """(bytes, note): HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KiB -> bytes).  The profile is REFUSED
    (bytes = None, the note says why) when it was collected for another workload size, for another kernel symbol than the
    one this run measures, or from kernel sources that differ from the ones the shipped library was built from."""
    path = PMC_TRAFFIC_JSON
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, 'no committed PMC profile (profiles/pmc_traffic.json)'
    if t.get('workload_elems') != n_elems:
        return None, 'committed PMC profile is for another workload size'
    if 'tq::fq_tensor<1,' not in str(t.get('kernel', '')):
        return None, 'committed PMC profile is for another kernel: ' + str(t.get('kernel'))[:80]
    try:
        now = kernel_source_hash()
    except OSError:
        now = None
    if t.get('kernel_source_sha256') != now:
        return None, ('STALE: committed PMC profile was collected from other kernel sources (%s..., now %s...) -- re-run '
                      'scripts/profile_gpu.sh' % (str(t.get('kernel_source_sha256'))[:12], str(now)[:12]))
    return t.get('traffic_bytes_per_launch'), (
        'committed profile, NOT a counter of this run: bytes per launch from profiles/pmc_traffic.json (rocprofv3 --pmc '
        'passes of this same command, gfx950 FETCH_SIZE x2 correction), kernel symbol and source hash match the built library')


def per_token_block(x, device, n_elems, B, S):
    """Per-token ranges (`--per-token`: axis = 1, reference main.py:359-376) and the dynamic step (`--dynamic`: estimate +
    quantize on EVERY call, quantization_manager.py:99-106) -- the one mode where the estimators are on the inference path."""
    from quantization.quantization_manager import QuantizationManager
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from utils.per_embd_quant_utils import set_act_quant_axis_and_groups

    def run(xin, dynamic, n_timed):
        m = QuantizationManager(qmethod=QMethods.asymmetric_uniform, init=RangeEstimators.current_minmax, qparams=dict(n_bits=8))
        set_act_quant_axis_and_groups(m, axis=1, n_groups=None)
        with torch.no_grad():
            m(xin)
            if not dynamic:
                m.fix_ranges()
            for _ in range(3):
                m(xin)
            _, ms = timed_region(lambda: m(xin), n_timed, False)
        return ms
    fixed_ms, dyn_ms = run(x, False, 20), run(x, True, 20)
    small = make_hidden(8, 128, device, seed=7)
    return {
        'shape': [B, S, 768], 'storage_dtype': 'bf16',
        'fixed_range': {'kernel_ms': round(fixed_ms, 4), 'GBps': round(n_elems * BYTES_PER_ELEM / fixed_ms / 1e6, 1),
                        'frac': round(n_elems * BYTES_PER_ELEM / fixed_ms / 1e6 / HBM_PEAK_GBS, 4), 'bytes_per_elem': BYTES_PER_ELEM},
        'dynamic': {'step_ms': round(dyn_ms, 4), 'GBps': round(n_elems * 6 / dyn_ms / 1e6, 1),
                    'frac': round(n_elems * 6 / dyn_ms / 1e6 / HBM_PEAK_GBS, 4), 'bytes_per_elem': 6,
                    'note': 'statistics + estimator + parameters + quantize per call: x is read twice (2 + 2 + 2 B per bf16 element)'},
        'dynamic_config_shape': {'shape': [8, 128, 768], 'call_us': round(run(small, True, 200) * 1e3, 2),
                                 'note': "ONE launch, one read of x (calib_rows_onepass_k: a token position's 8 x 768 values fit a "
                                         "block's registers); host-bound"},
    }


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
    (CPU test of `--gpus N`, tests/test_dist_gloo.py).  The line it prints is labelled and carries no measurement; it has
    the keys of the real line, and the raw communicator's agreement round (quantization/rccl.py `agree`) runs for real
    over the rendezvous store."""
    agreed = None
    if world > 1 or 'RANK' in os.environ:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')
        dist.barrier()
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t[0]) == world
        from quantization import rccl
        from torch.distributed import distributed_c10d as c10d
        agreed = rccl.agree(c10d._get_default_store(), 'tq_bench_dry_run', rank, world, True, timeout_s=60) == []
        dist.destroy_process_group()
    if rank == 0:
        blocks = {k: None for k in ('calibration', 'config_shape', 'calibration_model', 'adaround_dp', 'qat_dp', 'cpu_baseline',
                                    'roofline')}
        print(json.dumps({'metric': 'DRY RUN (no kernels launched, not a measurement)', 'value': None, 'dry_run': True,
                          'n_gpus': world, 'rccl_world_size': world, 'steps': args.steps, 'warmup': args.warmup,
                          'bring_up_agreement_round': agreed, **blocks}),
              flush=True)


# ---- optional blocks of the JSON line (after the headline measurement; a failure or a stall in any of them never
# ---- costs the headline: see _Watchdog) ---------------------------------------------------------------------------------
_REAL_STDOUT = None


def _claim_stdout():
    """The contract is ONE JSON line on stdout, but native libraries write to file descriptor 1 as well: RCCL prints a
    five-line version banner when a communicator is created (`RCCL version : ...`, seen in front of the line of every
    run that creates one).  From here on fd 1 is an alias of stderr for everybody, and the line goes out through the
    saved descriptor (`_emit`)."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def _emit(line):
    data = (line + '\n').encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line + '\n')
        sys.stdout.flush()
        return
    while data:
        data = data[os.write(_REAL_STDOUT, data):]


class _Watchdog:
    """The blocks below contain collectives that have never met more than one rank on this pool (raw RCCL bring-up, a
    captured ncclAllReduce, ...).  If they stall, every rank's timer fires: rank 0 prints the line it has -- headline
    complete, the unfinished block marked -- and all ranks leave through os._exit, so the driver's record survives."""

    def __init__(self, out, rank, seconds):
        import threading
        self.out, self.rank, self.stage = out, rank, 'start'
        self._done = threading.Event()
        self._t = threading.Thread(target=self._run, args=(float(seconds),), daemon=True)
        self._t.start()

    def _run(self, seconds):
        if self._done.wait(seconds):
            return
        if self.rank == 0:
            note = (f'optional block stalled in stage {self.stage!r} (watchdog, {seconds:.0f} s); headline fields are '
                    'complete')
            line = None
            for _ in range(5):                   # the main thread may be adding a key right now
                try:
                    snap = dict(self.out)
                    snap['incomplete'] = list(snap.get('incomplete', [])) + [note]
                    line = json.dumps(snap)
                    break
                except RuntimeError:
                    time.sleep(0.01)
            _emit(line if line is not None else json.dumps({'incomplete': [note]}))
        else:
            time.sleep(2.0)                  # let rank 0 print first
        os._exit(0)

    def cancel(self):
        self._done.set()


def _max_over_ranks(values, device, use_dist):
    if not use_dist:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def _wall_ms(fn, n, warm, use_dist):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def calibration_model_block(device, rank, world, use_dist, wd):
    """The latency-bound path of sharded calibration: a whole BERT-base calibrating forward (reference utils/utils.py:47-79
    is the loop, quantization_manager.py:99-106 is why every site's exchange is inline): 161 activation sites, each
    statistics -> all-reduce -> update -> quantize, through the harness model (harness/bert.py).

    * weak: every rank calibrates its own [8, 128] batch (per-GPU work fixed);
    * strong: a GLOBAL [128, 128] batch sharded over the ranks (16 samples per rank on 8 GPUs);
    each eager and -- on the raw-RCCL transport (or with no exchange at all) -- replayed as ONE hipGraph, collectives
    included.  Times are the MAX over ranks.  `exchange_latency_us`: the measured device time of one back-to-back
    all-reduce of 8 B / 6 KB (per-tensor / per-embedding statistics) / 103 KB (2-D MSE loss grid) / 9.4 MB (AdaRound
    gradient) on the transport in use."""
    from harness.bert import build_bert_base
    from quantization import distributed as tq_dist, options
    from quantization.graphs import GraphedForward
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    wd.stage = 'calibration_model: build'
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
              weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)
    model, _ = build_bert_base(seed=1000, **qp)
    model = model.to(device).eval()
    g = torch.Generator(device=device).manual_seed(4000)
    ids_global = torch.randint(1000, 30000, (128, 128), device=device, generator=g)      # the same on every rank
    g.manual_seed(5000 + rank)
    ids_weak = torch.randint(1000, 30000, (8, 128), device=device, generator=g)
    active = use_dist and tq_dist.is_enabled()
    raw = tq_dist.raw_comm()
    blk = {
        'model': 'BERT-base harness, W8A8 per-tensor, running min/max (BASELINE configs[1]), random init',
        'world_size': world,
        'transport': ('raw RCCL (tq_calibrate_minmax_rccl, one C call per site)' if raw is not None else
                      ('torch.distributed' if active else 'none (single GPU, no exchange)')),
    }
    with torch.no_grad():
        model.set_quant_state(True, True)
        model.estimate_ranges()
        legs = {'weak': ids_weak, 'strong': tq_dist.shard_batch(ids_global) if active else ids_global}
        for name, ids in legs.items():
            wd.stage = f'calibration_model: {name} eager'
            leg = {'per_rank_batch': list(ids.shape), 'global_batch': [ids.shape[0] * world if name == 'weak' else 128, 128]}
            from quantization.autoquant_utils import INT8_STATS
            i8_before = INT8_STATS['kernel_calls']
            model(ids)                   # untimed: the 102 weight quantizers estimate (and exchange) once, on the first forward
            # options.INT8_CALIBRATION (product default): Linears whose GEMM reaches INT8_CALIBRATION_MIN_MACS run as exact
            # integer GEMMs in a calibrating forward too -- none at [8,128], all 72 encoder Linears at [128,128] on one GPU
            leg['integer_gemm_linears_per_forward'] = INT8_STATS['kernel_calls'] - i8_before
            before = tq_dist.stats()
            n_eager = 10
            eager = _wall_ms(lambda: model(ids), n_eager, 2, use_dist)
            after = tq_dist.stats()
            if leg['integer_gemm_linears_per_forward']:
                wd.stage = f'calibration_model: {name} eager, fp32 GEMMs'
                keep_cal = options.INT8_CALIBRATION
                options.INT8_CALIBRATION = False
                try:
                    leg['eager_ms_fp32_gemms'] = round(_max_over_ranks([_wall_ms(lambda: model(ids), 5, 1, use_dist)], device, use_dist)[0], 4)
                finally:
                    options.INT8_CALIBRATION = keep_cal
            leg['collectives_per_forward'] = (after['minmax_calls'] + after['sum_calls'] - before['minmax_calls']
                                              - before['sum_calls']) / (n_eager + 2)
            leg['exchanged_bytes_per_forward'] = (after['bytes'] - before['bytes']) / (n_eager + 2)
            graph_ms = None
            if not active or raw is not None:          # c10d collectives cannot be captured (graphs.CaptureRefused)
                wd.stage = f'calibration_model: {name} hipGraph'
                try:
                    options.INPLACE_CALIBRATION_STATE = True
                    model(ids)
                    gf = GraphedForward(model, ids)
                    graph_ms = _wall_ms(lambda: gf(ids), 20, 3, use_dist)
                    del gf
                except Exception as e:       # noqa: BLE001
                    leg['hipgraph_error'] = repr(e)[:300]
                finally:
                    options.INPLACE_CALIBRATION_STATE = False
            eager, graph_ms = _max_over_ranks([eager, graph_ms if graph_ms is not None else -1.0], device, use_dist)
            leg['eager_ms'] = round(eager, 4)
            leg['hipgraph_ms'] = round(graph_ms, 4) if graph_ms >= 0 else None
            best = min(eager, graph_ms) if graph_ms >= 0 else eager
            leg['tokens_per_s'] = round(leg['global_batch'][0] * 128 / best * 1e3, 1)
            blk[name] = leg
        model.fix_ranges()
        # the calibrated model's fixed-range forward at the config shape (collective-free: configs[1]'s inference pass)
        wd.stage = 'calibration_model: fixed-range forward'
        # (both evaluation routes: the layered module chain -- what calibration above ran -- and the product's default for a
        # no-grad fixed-range forward, options.INT8_LINEAR = 'auto': exact-integer GEMMs + fused tails / attention core)
        from harness.routes import Route
        fx = {'per_rank_batch': [8, 128],
              'default_route': "options.INT8_LINEAR = 'auto' -> integer / fused ('fused_*' keys); 'layered_*' = options.INT8_LINEAR = False"}
        for tag, route in (('layered', 'layered'), ('fused', 'default')):
            with Route(model, route):
                e_ms = _wall_ms(lambda: model(ids_weak), 20, 3, use_dist)
                g_ms = None
                try:
                    with tq_dist.suspended():
                        gf = GraphedForward(model, ids_weak)
                    g_ms = _wall_ms(lambda: gf(ids_weak), 30, 3, use_dist)
                    del gf
                except Exception as e:       # noqa: BLE001
                    fx[tag + '_hipgraph_error'] = repr(e)[:300]
            e_ms, g_ms = _max_over_ranks([e_ms, g_ms if g_ms is not None else -1.0], device, use_dist)
            fx[tag + '_eager_ms'] = round(e_ms, 4)
            fx[tag + '_hipgraph_ms'] = round(g_ms, 4) if g_ms >= 0 else None
        blk['fixed_range_forward'] = fx
    # ---- per-exchange latency on the transport in use, measured (not assumed) -------------------------------------------
    wd.stage = 'calibration_model: exchange latency'
    lat = None
    if active:
        lat = {}
        for label, nbytes in (('8B', 8), ('6KB', 6144), ('103KB', 103424), ('9.4MB', 3072 * 768 * 4)):
            if raw is not None:
                us = raw.latency_us(nbytes)
            else:
                buf = torch.zeros(max(1, nbytes // 4), device=device)
                us = _wall_ms(lambda: dist.all_reduce(buf), 100, 10, True) * 1e3
            lat[label] = round(_max_over_ranks([us], device, use_dist)[0], 2)
    blk['exchange_latency_us'] = lat
    blk['exchange_latency_method'] = ('median over 20 groups of 10 back-to-back in-place fp32 SUM all-reduces, HIP events '
                                      'around each group, max over ranks' if raw is not None else
                                      ('wall clock of 100 back-to-back torch.distributed.all_reduce, max over ranks'
                                       if active else None))
    del model
    return blk


def adaround_dp_block(device, rank, world, use_dist, wd):
    """BASELINE configs[3]: data-parallel AdaRound of one BERT-base FFN layer, W4, weight [3072, 768]: every iteration draws
    a GLOBAL batch of 8 cached samples (128 tokens each), each rank evaluates its share and the 9.44 MB gradient dL/dW_q
    is SUM-all-reduced (quantization/adaround/adaround.py).  ms per iteration = (t(120 iterations) - t(20)) / 100, so
    the caching / initialisation cost cancels; max over ranks."""
    import copy
    from quantization import distributed as tq_dist
    from quantization.adaround import apply_adaround_to_layer
    from quantization.adaround.config import DEFAULT_ADAROUND_CONFIG
    from quantization.adaround.utils import AdaRoundMode, AdaRoundInitMode
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_model import QuantizedModel
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    from quantization import options
    wd.stage = 'adaround_dp'
    fin, fout, n, t = 768, 3072, 64, 128

    class Net(QuantizedModel):
        def __init__(self):
            super().__init__()
            self.fc = quantize_model(torch.nn.Linear(fin, fout), method=QMethods.symmetric_uniform, n_bits=4,
                                     weight_range_method=RangeEstimators.current_minmax)

        def forward(self, x):
            return self.fc(x)

    g = torch.Generator(device=device).manual_seed(1000)
    data = torch.randn(n, t, fin, generator=g, device=device)
    times = {}
    graph_was = options.GRAPH_ADAROUND
    try:
        for iters in (20, 20, 120):                     # the first run is the warm-up (allocator, first-touch, capture set-up)
            torch.manual_seed(1000)                     # same layer and same sample sequence on every rank
            net = Net().to(device).eval()
            net.set_quant_state(True, False)
            with torch.no_grad():
                net(data[:8])
            net.fix_ranges()
            cfg = copy.deepcopy(DEFAULT_ADAROUND_CONFIG)
            cfg.iters, cfg.round_mode, cfg.init = iters, AdaRoundMode.learned_hard_sigmoid, AdaRoundInitMode.range_estimator
            net.full_precision()
            net.fc.quantized_weights()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            t0 = time.perf_counter()
            apply_adaround_to_layer(net, net.fc, data, batch_size=8, act_quant=False, adaround_config=cfg)
            torch.cuda.synchronize()
            times[iters] = time.perf_counter() - t0
            del net
    finally:
        options.GRAPH_ADAROUND = graph_was
    ms = (times[120] - times[20]) / 100 * 1e3
    ms = _max_over_ranks([ms], device, use_dist)[0]
    active = use_dist and tq_dist.is_enabled()
    return {'layer': [fout, fin], 'bits': 4, 'global_batch': 8, 'tokens_per_sample': t,
            'per_rank_samples_per_iter': 8 / world if active else 8,
            'gradient_allreduce_bytes': fout * fin * 4 if active else 0,
            'ms_per_iter': round(ms, 4), 'iters_per_s': round(1e3 / ms, 1),
            'note': 'hipGraph replay of the loop is single-GPU only; with an active exchange the loop runs eagerly'}


def qat_dp_block(device, rank, world, use_dist, wd):
    """BASELINE configs[4] shape of work, at BERT-base scale: one data-parallel QAT step (forward, STE backward, bucketed
    gradient all-reduce of weights + learnable ranges overlapped with backward, SGD) of the 2-layer W8A8 harness model,
    per-rank batch [8, 128]; eager and as ONE hipGraph (quantization.graphs.GraphedTrainStep with grad_sync)."""
    from harness.bert import build_bert_base
    from quantization import distributed as tq_dist
    from quantization.data_parallel import GradientBuckets, train_step
    from quantization.graphs import GraphedTrainStep
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    wd.stage = 'qat_dp: build'
    qp = dict(method=QMethods.symmetric_uniform, act_method=QMethods.asymmetric_uniform, n_bits=8, n_bits_act=8,
              weight_range_method=RangeEstimators.current_minmax, act_range_method=RangeEstimators.running_minmax)
    model, _ = build_bert_base(seed=1000, num_layers=2, **qp)
    model = model.to(device)
    g = torch.Generator(device=device).manual_seed(6000 + rank)
    ids = torch.randint(1000, 30000, (8, 128), device=device, generator=g)
    labels = torch.randint(0, 2, (8,), device=device, generator=g)
    with torch.no_grad(), tq_dist.suspended():
        model.eval()
        model.set_quant_state(True, True)
        model.estimate_ranges()
        model(ids)
    model.learn_ranges()
    model.set_quant_state(True, True)
    model.train()
    if use_dist and tq_dist.is_enabled():
        from quantization.data_parallel import broadcast_parameters
        broadcast_parameters(model)          # every rank calibrated on its own batch: replicas start from rank 0's ranges
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-4)
    gb = GradientBuckets(params)
    loss_fn = torch.nn.functional.cross_entropy
    active = use_dist and tq_dist.is_enabled()
    wd.stage = 'qat_dp: eager'
    eager = _wall_ms(lambda: train_step(model, loss_fn, opt, gb, (ids,), (labels,)), 10, 3, use_dist)
    graph_ms, err = None, None
    if not active or tq_dist.raw_comm() is not None:
        wd.stage = 'qat_dp: hipGraph'
        try:
            step = GraphedTrainStep(model, loss_fn, opt, (ids,), (labels,), grad_sync=gb)
            graph_ms = _wall_ms(lambda: step((ids,), (labels,)), 20, 3, use_dist)
        except Exception as e:       # noqa: BLE001
            err = repr(e)[:300]
    eager, graph_ms = _max_over_ranks([eager, graph_ms if graph_ms is not None else -1.0], device, use_dist)
    blk = {'model': 'BERT harness, 2 encoder layers, W8A8, learnable ranges, SGD', 'per_rank_batch': [8, 128],
           'trainable_tensors': len(params), 'gradient_bytes': sum(gb.bucket_sizes()), 'buckets': gb.n_buckets,
           'allreduces_per_step': gb.n_buckets if active else 0,
           'eager_ms': round(eager, 4), 'hipgraph_ms': round(graph_ms, 4) if graph_ms >= 0 else None,
           'samples_per_s': round(8 * world / (min(eager, graph_ms) if graph_ms >= 0 else eager) * 1e3, 1)}
    if err:
        blk['hipgraph_error'] = err
    gb.close()
    return blk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--seq', type=int, default=512)
    ap.add_argument('--sweep', action='store_true', help='also time the SURVEY.md 8d shape sweep')
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--headline-only', action='store_true',
                    help='skip the whole-model calibration / AdaRound / QAT blocks and the CPU baseline')
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
    _claim_stdout()
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

    B, S = args.batch, args.seq
    x = make_hidden(B, S, device, seed=1000 + rank)
    n_elems = x.numel()

    def new_quantizer():
        q = QuantizedActivation(act_method=QMethods.asymmetric_uniform, n_bits_act=8,
                                act_range_method=RangeEstimators.running_minmax).to(device)
        q.quantized_acts()
        q.eval()
        return q

    # ---- the hot path: fixed-range fake-quant forward -----------------------------------------------------------------
    # Measured FIRST and with no statistics exchange configured: every rank calibrates on its own two batches, fixes the
    # range and runs the collective-free forward on its own shard (weak scaling).  Nothing that has not yet run on more
    # than one rank on this pool (raw communicator bring-up, captured collectives) can cost the headline number.
    qa = new_quantizer()
    for xb in (x, make_hidden(B, S, device, seed=2000 + rank)):
        qa(xb)
    qa.activation_quantizer.fix_ranges()
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
    wall = _max_over_ranks([wall], device, use_dist)[0]

    total_elems = n_elems * world * args.steps
    value = total_elems / wall / 1e6
    kernel_s = ev_ms / 1e3
    achieved = n_elems * BYTES_PER_ELEM / kernel_s / 1e9

    out = {
        'metric': 'M elems/sec fake-quant fwd (BERT-base act tensor); achieved HBM GB/s vs peak',
        'value': round(value, 1),
        'unit': 'M elems/s',
        'n_gpus': world,
        'rccl_world_size': None,           # filled from the communicator below
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
            'traffic': pmc_traffic(n_elems)[0],
            'traffic_source': pmc_traffic(n_elems)[1],
            'kernel': 'tq::fq_tensor<bf16>',
            'kernel_ms': round(ev_ms, 4),
            'algorithmic_bytes_per_launch': n_elems * BYTES_PER_ELEM,
        },
        'cpu_baseline': None,
    }

    # the config shape [8, 128, 768] ("report both"): one launch per activation site of a B=8 forward --
    # launch-latency bound, 3 MB of traffic
    xs = make_hidden(8, 128, device, seed=7)
    with torch.no_grad():
        for _ in range(10):
            qa(xs)
        _, ms_small = timed_region(lambda: qa(xs), 200, False)
    out['config_shape'] = {'shape': [8, 128, 768], 'kernel_us': round(ms_small * 1e3, 3),
                           'M_elems_s': round(xs.numel() / ms_small / 1e3, 1),
                           'GBps': round(xs.numel() * BYTES_PER_ELEM / ms_small / 1e6, 1),
                           'note': 'HIP-event span of 200 back-to-back calls / 200 (launch boundaries included)'}
    # ... and the same 161 sites as ONE multi-tensor call (tq_fake_quant_multi_fwd: 40 tensors per launch): what independent
    # sites cost when they do not pay a launch each (a forward's sites depend on each other; a model's 102 weight tensors
    # do not -- quantization.autoquant_utils.prequantize_weights)
    try:
        from quantization import _hip as _tq_hip
        q_small = qa.activation_quantizer.quantizer
        sites = [xs.clone() for _ in range(161)]
        plan = _tq_hip.backend().fake_quant_multi_plan(
            [(t, q_small._delta, q_small._zero_float, None, q_small.n_bits, False, False, q_small.eps, 1, 1) for t in sites])
        for _ in range(3):
            _tq_hip.backend().fake_quant_multi_launch(plan)
        _, ms_multi = timed_region(lambda: _tq_hip.backend().fake_quant_multi_launch(plan), 20, False)
        out['config_shape']['batched_161_sites'] = {
            'us_per_site': round(ms_multi * 1e3 / 161, 3), 'ms_per_call': round(ms_multi, 4),
            'GBps': round(161 * xs.numel() * BYTES_PER_ELEM / ms_multi / 1e6, 1),
            'note': '161 independent [8,128,768] tensors, one C call = 5 launches of <= 40 tensors'}
        del sites, plan
    except Exception as e:       # noqa: BLE001 -- optional figure
        out['config_shape']['batched_161_sites'] = {'error': repr(e)[:300]}

    if not args.headline_only:
        try:
            out['per_token'] = per_token_block(x, device, n_elems, B, S)
        except Exception as e:       # noqa: BLE001 -- optional figures
            out['per_token'] = {'error': repr(e)[:300]}

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
    del xs

    # ---- everything below runs under the watchdog -----------------------------------------------------------------------
    wd = _Watchdog(out, rank, float(os.environ.get('TQ_BENCH_WATCHDOG_S', '300')))

    # Everything from here on is optional: an exception is recorded in the line instead of costing the headline.
    try:
        # statistics exchange: the raw RCCL communicator inside libtq_hip.so when the backend is `nccl` (quantization/rccl.py:
        # two-phase bring-up agreed through the rendezvous store, self-tested); torch.distributed is the fallback
        transport = 'none'
        if use_dist:
            wd.stage = 'exchange bring-up'
            try:
                tq_dist.enable(force=(world == 1))
            except Exception as e:       # noqa: BLE001
                print(f'[bench] raw RCCL exchange unavailable ({e!r}): statistics go through torch.distributed', file=sys.stderr)
                tq_dist.enable(force=(world == 1), raw=False)
            raw = tq_dist.raw_comm()
            transport = 'raw RCCL (tq_calibrate_minmax_rccl)' if raw is not None else f'torch.distributed ({backend})'
            # the size of the communicator the data path actually uses (ncclCommCount), not the launcher's environment
            out['rccl_world_size'] = raw.rank_world()[1] if raw is not None else dist.get_world_size()
            assert out['rccl_world_size'] == world, f'communicator spans {out["rccl_world_size"]} ranks, WORLD_SIZE is {world}'
        else:
            out['rccl_world_size'] = 1

        # ---- calibration of ONE quantizer on the large tensor: estimate + quantize, one fused MAX all-reduce per step ------
        wd.stage = 'calibration (one quantizer)'
        qc = new_quantizer()
        for xb in (x, make_hidden(B, S, device, seed=2000 + rank)):
            qc(xb)
        cal_steps = max(4, min(args.steps, 20))
        cal_wall, _ = timed_region(lambda: qc(x), cal_steps, use_dist)
        # --mailbox / TQ_BENCH_MAILBOX=1: the same step with the statistics exchanged through the P2P mailbox kernel instead
        # of ncclAllReduce (reported next to the RCCL figure, never instead of it; skipped if the set-up or the self-test
        # against RCCL fails).  Opt-in: validated with two processes on one device and with a 1-rank RCCL group only.
        mail_wall = None
        if use_dist and backend == 'nccl' and (args.mailbox or os.environ.get('TQ_BENCH_MAILBOX', '0') == '1'):
            wd.stage = 'calibration (mailbox)'
            try:
                tq_dist.enable(force=(world == 1), mailbox=True, raw=tq_dist.raw_comm() is not None)
                if tq_dist.mailbox_active():
                    qc(x)
                    mail_wall, _ = timed_region(lambda: qc(x), cal_steps, use_dist)
            except Exception as e:       # noqa: BLE001
                print(f'[bench] mailbox leg skipped: {e!r}', file=sys.stderr)
            finally:
                tq_dist.enable(force=(world == 1), mailbox=False, raw=tq_dist.raw_comm() is not None)
        cal_wall, mw = _max_over_ranks([cal_wall, mail_wall if mail_wall is not None else -1.0], device, use_dist)
        mail_wall = mw if mail_wall is not None else None
        out['calibration'] = {
            'what': 'estimate (tq_minmax -> range_update -> set_range) + quantize per step; '
                    + ('one fused MAX all-reduce of [-min;max] per step over RCCL' if world > 1
                       else ('1-rank communicator, collectives forced on' if use_dist else 'single GPU, no collective')),
            'transport': transport,
            'value': round(n_elems * world * cal_steps / cal_wall / 1e6, 1),
            'unit': 'M elems/s',
            'ms_per_step': round(cal_wall / cal_steps * 1e3, 4),
        }
        if mail_wall is not None:
            out['calibration']['p2p_mailbox'] = {
                'what': 'same step, [-min;max] exchanged by the P2P mailbox kernel (tq_mailbox_allreduce_max) instead of RCCL',
                'value': round(n_elems * world * cal_steps / mail_wall / 1e6, 1), 'unit': 'M elems/s',
                'ms_per_step': round(mail_wall / cal_steps * 1e3, 4)}
        del x, qc, qa
    except Exception as e:       # noqa: BLE001
        out.setdefault('incomplete', []).append(f'stage {wd.stage!r} failed: {e!r}'[:600])
        print(f'[bench] rank {rank}: stage {wd.stage!r} failed: {e!r}', file=sys.stderr)
        if out.get('rccl_world_size') is None:
            out['rccl_world_size'] = dist.get_world_size() if use_dist else 1

    # ---- whole-model sharded calibration, data-parallel AdaRound / QAT steps, CPU baseline ------------------------------
    if not args.headline_only:
        for key, fn in (('calibration_model', calibration_model_block), ('adaround_dp', adaround_dp_block),
                        ('qat_dp', qat_dp_block)):
            torch.cuda.empty_cache()
            try:
                out[key] = fn(device, rank, world, use_dist, wd)
            except Exception as e:       # noqa: BLE001 -- with N > 1 a one-rank failure may stall the peers: the watchdog ends it
                out[key] = {'error': repr(e)[:500]}
                print(f'[bench] rank {rank}: block {key} failed: {e!r}', file=sys.stderr)
        wd.stage = 'cpu_baseline'
        if rank == 0 and world == 1 and not args.no_cpu:
            out['cpu_baseline'] = cpu_baseline()
    wd.cancel()

    if rank == 0:
        _emit(json.dumps(out))
    if use_dist:
        try:
            tq_dist.disable()
        finally:
            dist.destroy_process_group()


if __name__ == '__main__':
    main()
