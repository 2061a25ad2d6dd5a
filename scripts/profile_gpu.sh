#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + separate PMC passes for the bench workload.
# Outputs land in gpurun_out/prof_<tag>/ ; summarise with scripts/summarize_profiles.py.
set -u
TAG=${1:-r01}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --headline-only"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/bench_trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- $BENCH > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- $BENCH > "$OUT/bench_write.log" 2>&1
find "$OUT" -name "*.csv" | head -30
python $ROOT/scripts/summarize_profiles.py "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep only small artefacts (csv stats + summaries)
find "$OUT" -name "*.db" -delete 2>/dev/null
du -sh "$OUT"
