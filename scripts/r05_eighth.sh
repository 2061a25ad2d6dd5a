#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd "$ROOT" || exit 1
O=$ROOT/gpurun_out/r05
mkdir -p "$O"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_per_token.py -q -m gpu -x > "$O/tests8.log" 2>&1; echo "tests rc=$?"
grep -v amdgpu.ids "$O/tests8.log" | tail -8
for rep in 1 2; do
timeout 300 python scripts/kernel_bench.py --only fq 2>&1 | grep -v amdgpu.ids | grep "K2r\|dynamic per-token\|K1 "
TQ_ROWS_SEQ_MIN_MB=100000 timeout 300 python scripts/kernel_bench.py --only fq 2>&1 | grep -v amdgpu.ids | grep "K2r\|dynamic per-token" | sed 's/^/  [wave kernel] /'
done
