import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel ms per forward:", tot / n / 1e6, " kernels per forward:", sum(int(r["Calls"]) for r in rows) / n)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print("%5.1f%%  calls/fwd %7.1f  avg %7.2f us  %s" % (float(r["TotalDurationNs"]) / tot * 100, int(r["Calls"]) / n,
                                                        float(r["AverageNs"]) / 1e3, r["Name"][:120]))
