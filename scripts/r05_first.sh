#!/bin/bash
# Round-5 first contact (run ON THE GPU BOX via gpurun): MX-MFMA probe, per-token kernels, integer-vs-reference experiment.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd "$ROOT" || exit 1
O=$ROOT/gpurun_out/r05
mkdir -p "$O"
export TMPDIR=/tmp
timeout 120 tools/tuning/mx_probe > "$O/mx_probe.txt" 2>&1; echo "mx_probe rc=$?"
timeout 600 python -m pytest tests/test_per_token.py -q -m gpu -x > "$O/per_token_tests.log" 2>&1; echo "per-token tests rc=$?"
tail -5 "$O/per_token_tests.log"
timeout 300 python scripts/kernel_bench.py --only fq 2>&1 | grep -v amdgpu.ids > "$O/kernel_bench_fq.txt"; echo "kernel_bench rc=$?"
timeout 900 python scripts/int_vs_reference.py > "$O/int_vs_reference.json" 2> "$O/int_vs_reference.err"; echo "int_vs_reference rc=$?"
tail -5 "$O/int_vs_reference.err"
cat "$O/mx_probe.txt"
cat "$O/kernel_bench_fq.txt"
