#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd "$ROOT" || exit 1
O=$ROOT/gpurun_out/r05
mkdir -p "$O"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_validate_cli.py tests/test_calibration_graph.py -q -m gpu -x > "$O/tests10.log" 2>&1; echo "tests rc=$?"
grep -v amdgpu.ids "$O/tests10.log" | tail -8
python scripts/config_bench.py > "$O/config_bench.json" 2> "$O/config_bench.err"; echo "config_bench rc=$?"
tail -3 "$O/config_bench.err"
python - <<'PY'
import json
c=json.load(open('gpurun_out/r05/config_bench.json'))
print(json.dumps(c.get('bert_base_dynamic_per_token_b8_t128'),indent=1))
PY
