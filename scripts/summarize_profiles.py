#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel stats + PMC passes) into profiles/-sized summaries."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict



def kernel_table(out_dir, dest_dir):
    """`summarize_profiles.py kernel-table OUT DEST`: join scripts/kernel_bench.py's manifest (algorithmic bytes /
    ops per launch) with rocprofv3's per-dispatch kernel trace of the same run -> DEST/kernel_table.md
    + kernel_table.json: avg ns -> achieved -> fraction of the bounding peak, per kernel and shape."""
    man = json.load(open(os.path.join(out_dir, 'manifest.json')))
    traces = glob.glob(os.path.join(out_dir, '**', '*kernel_trace.csv'), recursive=True)
    if not traces:
        raise SystemExit('no *kernel_trace.csv under ' + out_dir)
    disp = []
    for r in csv.DictReader(open(traces[0])):
        disp.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    disp.sort()
    cursor = defaultdict(int)       # per pattern: how many matching dispatches earlier manifest rows consumed

    def patterns_of(row):
        return row.get('patterns') or [row['pattern']]

    rows = []
    for m in man:
        # each run() = 3 warm + reps timed launches of its kernel(s); rows sharing a pattern follow each other in time.
        # A composite row (`patterns`: several kernels per call, e.g. the dynamic estimate + quantize step) sums the
        # averages of its kernels.
        total_avg, total_min, total_max, n_disp, outliers_all, names = 0.0, 0, 0, None, 0, []
        for pat in patterns_of(m):
            match = [d for d in disp if pat in d[2]]
            same = [x for x in man if pat in patterns_of(x)]
            per_run = len(match) // len(same) if same else 0
            mine = match[cursor[pat]:cursor[pat] + per_run]
            cursor[pat] += per_run
            timed = mine[3:] if len(mine) > 3 else mine
            if not timed:
                n_disp = 0
                break
            ns = [e - s for s, e, _ in timed]
            # a dispatch that is pre-empted (another client of the box, a clock transition) shows up as one launch of
            # 10-20x the others; such outliers (> 3x the median) are left out of the average and counted in the table
            med = sorted(ns)[len(ns) // 2]
            keep = [v for v in ns if v <= 3 * med]
            outliers_all += len(ns) - len(keep)
            total_avg += sum(keep) / len(keep)
            total_min += min(keep)
            total_max += max(keep)
            n_disp = len(timed)
            names.append(timed[0][2].split('(')[0][:80])
        if not n_disp:
            continue
        ach = m['unit_per_launch'] / total_avg          # bytes/ns = GB/s ; ops/ns = GOP/s
        rows.append({**m, 'kernel_name': ' + '.join(names), 'dispatches': n_disp,
                     'rocprof_avg_ns': round(total_avg, 1), 'rocprof_min_ns': total_min, 'rocprof_max_ns': total_max,
                     'outliers_excluded': outliers_all,
                     'achieved': round(ach, 1), 'frac': round(ach / m['peak'], 4)})
    os.makedirs(dest_dir, exist_ok=True)
    json.dump(rows, open(os.path.join(dest_dir, 'kernel_table.json'), 'w'), indent=1)
    with open(os.path.join(dest_dir, 'kernel_table.md'), 'w') as f:
        f.write('| family | what | kernel (rocprofv3) | launches | avg ns | min ns | algorithmic bytes / ops per launch | '
                'achieved | peak | frac | HIP-event us (same run) |\n|---|---|---|---|---|---|---|---|---|---|---|\n')
        for r in rows:
            f.write(f"| {r['family']} | {r['label']} | `{r['kernel_name']}` | {r['dispatches']}"
                    f"{' (-%d outlier)' % r['outliers_excluded'] if r['outliers_excluded'] else ''} | {r['rocprof_avg_ns']:.0f} | "
                    f"{r['rocprof_min_ns']} | {r['unit_per_launch']:.4g} {r['unit']} | {r['achieved']:.0f} {r['peak_unit']} | "
                    f"{r['peak']:.0f} | **{100 * r['frac']:.1f} %** | {r['event_us']} |\n")
    for r in rows:
        print(f"{r['label'][:50]:50s} {r['rocprof_avg_ns']:10.0f} ns  {r['achieved']:9.0f} {r['peak_unit']}  {100 * r['frac']:5.1f} %")


if len(sys.argv) > 1 and sys.argv[1] == 'kernel-table':
    kernel_table(sys.argv[2], sys.argv[3])
    sys.exit(0)

out, tag = sys.argv[1], sys.argv[2]


def find(pattern):
    hits = glob.glob(os.path.join(out, '**', pattern), recursive=True)
    return hits[0] if hits else None


summary = {'tag': tag}
stats = find('*kernel_stats.csv')
if stats:
    rows = list(csv.DictReader(open(stats)))
    summary['kernel_stats'] = [
        {k: r[k] for k in r if k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage',
                                      'MinNs', 'MaxNs', 'StdDev')} for r in rows[:25]]
    print('== kernel stats (top 15) ==')
    for r in rows[:15]:
        print(f"{r.get('Name','')[:90]:90s} calls={r.get('Calls')} avg_ns={r.get('AverageNs')} pct={r.get('Percentage')}")

for key, counter in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    f = find(f'*{key.split("_")[1]}*counter_collection.csv') or find('*counter_collection.csv')
    path = None
    for cand in glob.glob(os.path.join(out, key, '**', '*counter_collection.csv'), recursive=True):
        path = cand
    if not path:
        continue
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r.get('Counter_Name') != counter:
            continue
        name = r.get('Kernel_Name', '')
        agg[name][0] += 1
        agg[name][1] += float(r.get('Counter_Value', 0))
    summary[counter] = {n: {'dispatches': c, 'mean_value': v / max(c, 1)} for n, (c, v) in agg.items()}
    print(f'== {counter} (mean per dispatch, raw counter units = KiB) ==')
    for n, (c, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
        print(f'{n[:90]:90s} n={c} mean={v / max(c, 1):.1f}')

json.dump(summary, open(os.path.join(out, f'summary_{tag}.json'), 'w'), indent=1)

# bench.py reads profiles/pmc_traffic.json for `roofline.traffic`: bytes per launch of the dominant kernel
# (tq::fq_tensor<bf16>, streaming variant) = 2 x FETCH_SIZE + WRITE_SIZE, counters in KiB (gfx950 counts 64 B per
# 128-byte request on wide coalesced read streams -- MI355X_MICROARCH.md, HBM section).
if 'FETCH_SIZE' in summary and 'WRITE_SIZE' in summary:
    pick = [n for n in summary['FETCH_SIZE'] if n.startswith('void tq::fq_tensor<1,') and n in summary['WRITE_SIZE']]
    if pick:
        name = max(pick, key=lambda n: summary['FETCH_SIZE'][n]['dispatches'])
        fetch, write = summary['FETCH_SIZE'][name]['mean_value'], summary['WRITE_SIZE'][name]['mean_value']
        import hashlib
        h = hashlib.sha256()
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for rel in ('csrc/tq_fake_quant.hip', 'csrc/tq_device.h', 'csrc/tq_host.h'):      # == bench.py KERNEL_SOURCES
            h.update(open(os.path.join(root, 'transformer-quantization_amd', rel), 'rb').read())
        traffic = {
            'workload_elems': 1024 * 512 * 768, 'kernel': name, 'kernel_source_sha256': h.hexdigest(),
            'fetch_size_kib_mean': fetch, 'write_size_kib_mean': write,
            'correction': 'gfx950: FETCH_SIZE counts 64 B per 128 B request for wide coalesced streams -> x2 '
                          '(MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported; units KiB',
            'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over '
                      f'`python bench.py --steps 20 --warmup 5 --no-cpu`, {tag} (scripts/profile_gpu.sh)',
            'traffic_bytes_per_launch': int(round((2 * fetch + write) * 1024)),
        }
        json.dump(traffic, open(os.path.join(out, 'pmc_traffic.json'), 'w'), indent=1)
        print('== traffic per launch:', traffic['traffic_bytes_per_launch'], 'bytes (algorithmic',
              traffic['workload_elems'] * 4, ')')
