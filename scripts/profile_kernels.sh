#!/bin/bash
# Run ON THE GPU BOX (via gpurun): clock-settled per-family kernel trace -> gpurun_out/prof_kernels_<tag>/ ;
# then `python scripts/summarize_profiles.py kernel-table gpurun_out/prof_kernels_<tag> profiles/<tag>` here.
set -u
TAG=${1:-r02}
ONLY=${2:-}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_kernels_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- \
  python $ROOT/scripts/kernel_bench.py --manifest "$OUT/manifest.json" ${ONLY:+--only $ONLY} > "$OUT/kernel_bench.log" 2>&1
cat "$OUT/kernel_bench.log" | grep -v amdgpu.ids
python $ROOT/scripts/summarize_profiles.py kernel-table "$OUT" "$OUT/table" > "$OUT/table.log" 2>&1
tail -5 "$OUT/table.log"
find "$OUT" -name "*.db" -delete 2>/dev/null
# the per-dispatch trace is large; keep the stats + a compressed trace
find "$OUT" -name "*kernel_trace.csv" -exec gzip -f {} \;
du -sh "$OUT"
