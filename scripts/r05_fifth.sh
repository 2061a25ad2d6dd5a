#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd "$ROOT" || exit 1
O=$ROOT/gpurun_out/r05
mkdir -p "$O"
export TMPDIR=/tmp
L=$ROOT/transformer-quantization_amd/lib
{
for rep in 1 2; do
  for v in base ln_w5 ln_w6; do
    echo "== $v (rep $rep)"
    if [ $v = base ]; then unset TQ_LIB_PATH; else export TQ_LIB_PATH=$L/libtq_$v.so; fi
    timeout 300 python scripts/kernel_bench.py --only tails 2>&1 | grep -v amdgpu.ids | grep "LayerNorm tail"
  done
done
} > "$O/ln_occupancy_ab.txt" 2>&1
unset TQ_LIB_PATH
cat "$O/ln_occupancy_ab.txt"
timeout 1500 python -m pytest tests -q -m gpu -rs > "$O/gpu_tests_full_suite.log" 2>&1; echo "suite rc=$?"
tail -8 "$O/gpu_tests_full_suite.log"
