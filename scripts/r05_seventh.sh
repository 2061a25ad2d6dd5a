#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd "$ROOT" || exit 1
O=$ROOT/gpurun_out/r05
mkdir -p "$O"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_validate_cli.py tests/test_hip_parity.py tests/test_fuzz_parity.py tests/test_calibration_graph.py -q -m gpu -x > "$O/tests7.log" 2>&1; echo "tests rc=$?"
grep -v amdgpu.ids "$O/tests7.log" | tail -8
timeout 300 python tools/tuning/fast_eager_prof.py 2>&1 | grep -v amdgpu > "$O/fast_eager_prof.txt"
head -120 "$O/fast_eager_prof.txt"
