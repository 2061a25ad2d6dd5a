"""Top kernels of a rocprofv3 --kernel-trace --stats output directory:  python tools/tuning/stats_top.py DIR [n_forwards]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time per forward: %.3f ms' % (tot / n / 1e6))
for r in rows[:24]:
    print('%-100s calls=%6s avg_us=%9.1f total_ms_per_fwd=%7.3f pct=%s' % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3,
                                                                        float(r['TotalDurationNs']) / n / 1e6, r['Percentage']))
