#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd "$ROOT" || exit 1
O=$ROOT/gpurun_out/r05
mkdir -p "$O"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_ln.py tests/test_lockstep.py tests/test_bert_e2e.py tests/test_mobilebert_e2e.py tests/test_linear_i8.py -q -m gpu -x > "$O/tests4.log" 2>&1; echo "tests rc=$?"
grep -v amdgpu.ids "$O/tests4.log" | tail -25
timeout 600 python bench.py > "$O/bench_n1.json" 2> "$O/bench_n1.err"; echo "bench rc=$?"
tail -3 "$O/bench_n1.err"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_n1.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'])
print(json.dumps(d.get('calibration_model',{}).get('fixed_range_forward'),indent=1))
print({k:(v.get('eager_ms'),v.get('hipgraph_ms')) for k,v in d.get('calibration_model',{}).items() if isinstance(v,dict) and 'eager_ms' in v})
PY
