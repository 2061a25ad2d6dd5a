#!/usr/bin/env python3
"""Row-by-row comparison of two kernel tables (profiles/rNN/kernel_table.json): which kernel families got slower or faster
by more than the box-to-box spread.  Written in round 6 after a shared-header change made ONE tile kernel 2.5x slower and
only its table row showed it.

    python scripts/compare_kernel_tables.py profiles/r05/kernel_table.json profiles/r06/kernel_table.json [--md OUT.md]
"""
import json
import sys

THRESH = 0.08      # boxes differ by +-5 %


def load(path):
    # (round 6 renamed the per-token row with its kernel: fq_rows_wave -> fq_rows_tab)
    return {r['label'].replace('fq_rows_wave', 'fq_rows_tab'): r for r in json.load(open(path))}


def main():
    a_path, b_path = sys.argv[1], sys.argv[2]
    a, b = load(a_path), load(b_path)
    rows = []
    for label, rb in b.items():
        ra = a.get(label)
        if ra is None:
            rows.append((label, None, rb['rocprof_avg_ns'], None, 'new row'))
            continue
        ta, tb = ra['rocprof_avg_ns'], rb['rocprof_avg_ns']
        rel = tb / ta - 1.0
        rows.append((label, ta, tb, rel, 'SLOWER' if rel > THRESH else ('faster' if rel < -THRESH else '')))
    lines = [f'| kernel-table row | {a_path} avg us | {b_path} avg us | change | |', '|---|---|---|---|---|']
    for label, ta, tb, rel, tag in sorted(rows, key=lambda r: -(r[3] if r[3] is not None else -9)):
        if tag:
            lines.append(f"| {label} | {'' if ta is None else '%.1f' % (ta / 1e3)} | {tb / 1e3:.1f} | "
                         f"{'' if rel is None else '%+.0f %%' % (100 * rel)} | {tag} |")
    n_same = sum(1 for r in rows if not r[4])
    lines.append(f'\n{n_same} of {len(rows)} rows within +-{int(100 * THRESH)} % (the box-to-box spread); '
                 f'{sum(1 for r in rows if r[4] == "SLOWER")} slower, {sum(1 for r in rows if r[4] == "faster")} faster.')
    text = '\n'.join(lines)
    print(text)
    if '--md' in sys.argv:
        open(sys.argv[sys.argv.index('--md') + 1], 'w').write(text + '\n')
    return 0


if __name__ == '__main__':
    sys.exit(main())
