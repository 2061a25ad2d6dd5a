#!/bin/bash
# Round-6 GPU call 2: the reworked attention core (tests first), its timing and counters, then the whole GPU suite with the
# second pass under the default route.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd "$ROOT" || exit 1
O=$ROOT/gpurun_out/r06b
mkdir -p "$O"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_attention_i8.py tests/test_int_oracle.py tests/test_fused_ln.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -15
python tools/tuning/attn_graph.py 2>&1 | grep -v amdgpu.ids | tee "$O/attn_graph.txt"
bash scripts/pmc_attention.sh 64 > "$O/pmc64.log" 2>&1
bash scripts/pmc_attention.sh 8 > "$O/pmc8.log" 2>&1
cp gpurun_out/attention_pmc_B*.json "$O/"
grep -E "SQ_INSTS_VALU|SQ_WAVE_CYCLES|SQ_ACTIVE_INST_VALU|SQ_WAIT_INST_ANY|SQ_BUSY_CYCLES" "$O"/attention_pmc_B64.json "$O"/attention_pmc_B8.json
timeout 1500 python -m pytest tests -q -m gpu -rf 2>&1 | grep -v amdgpu.ids > "$O/gpu_suite.log"
tail -25 "$O/gpu_suite.log"
