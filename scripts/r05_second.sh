#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd "$ROOT" || exit 1
O=$ROOT/gpurun_out/r05
mkdir -p "$O"
export TMPDIR=/tmp
timeout 900 python scripts/int_vs_reference.py > "$O/int_vs_reference.json" 2> "$O/int_vs_reference.err"; echo "int_vs_reference rc=$?"
tail -5 "$O/int_vs_reference.err"
timeout 120 tools/tuning/mfma_valu_overlap > "$O/mfma_valu_overlap_r04_binary.txt" 2>&1; echo "overlap rc=$?"
head -8 "$O/mfma_valu_overlap_r04_binary.txt"
