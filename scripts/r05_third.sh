#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd "$ROOT" || exit 1
O=$ROOT/gpurun_out/r05
mkdir -p "$O"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lockstep.py tests/test_bert_e2e.py tests/test_mobilebert_e2e.py -q -m gpu -s -k "lockstep or readme or default_route" > "$O/new_tests.log" 2>&1; echo "new tests rc=$?"
grep -v amdgpu.ids "$O/new_tests.log" | tail -40
timeout 900 python scripts/int_vs_reference.py > "$O/int_vs_reference.json" 2> "$O/int_vs_reference.err"; echo "int_vs_reference rc=$?"
tail -5 "$O/int_vs_reference.err"
timeout 1500 python -m pytest tests -q -m gpu -rs > "$O/gpu_tests_full_suite.log" 2>&1; echo "suite rc=$?"
tail -30 "$O/gpu_tests_full_suite.log"
