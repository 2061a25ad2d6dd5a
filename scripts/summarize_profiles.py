#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel stats + PMC passes) into profiles/-sized summaries."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

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
