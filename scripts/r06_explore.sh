#!/bin/bash
# Round-6 GPU exploration call: the GPU suite under the product's default route (which tests are route-sensitive?),
# the attention core's phase profile and counters.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd "$ROOT" || exit 1
O=$ROOT/gpurun_out/r06x
mkdir -p "$O"
export TMPDIR=/tmp
TQ_TEST_ROUTE=default timeout 1200 python -m pytest tests -q -m gpu -rf 2>&1 | grep -v amdgpu.ids > "$O/gpu_suite_default_route.log"
tail -60 "$O/gpu_suite_default_route.log"
python tools/tuning/attn_prof.py 2>&1 | grep -v amdgpu.ids > "$O/attn_prof.txt"
python tools/tuning/attn_graph.py 2>&1 | grep -v amdgpu.ids > "$O/attn_graph.txt"
cat "$O/attn_prof.txt" "$O/attn_graph.txt"
bash scripts/pmc_attention.sh 64 > "$O/pmc64.log" 2>&1
bash scripts/pmc_attention.sh 8 > "$O/pmc8.log" 2>&1
cp gpurun_out/attention_pmc_B*.json "$O/"
cat "$O"/attention_pmc_B64.json "$O"/attention_pmc_B8.json
