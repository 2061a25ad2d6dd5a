#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd "$ROOT" || exit 1
O=$ROOT/gpurun_out/r05
mkdir -p "$O"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_per_token.py tests/test_validate_cli.py tests/test_hip_parity.py -q -m gpu -x > "$O/tests6.log" 2>&1; echo "tests rc=$?"
grep -v amdgpu.ids "$O/tests6.log" | tail -15
python tools/tuning/py_overhead2.py 2>&1 | grep -v amdgpu
timeout 300 python scripts/kernel_bench.py --only dyn_small 2>&1 | grep -v amdgpu.ids
python - <<'PY'
import sys, time
sys.path[:0]=['transformer-quantization_amd','.']
import torch
from quantization.quantization_manager import QuantizationManager
from quantization.quantizers import QMethods
from quantization.range_estimators import RangeEstimators
from utils.per_embd_quant_utils import set_act_quant_axis_and_groups
x=torch.randn(8,128,768,device='cuda')
mgr = QuantizationManager(qmethod=QMethods.asymmetric_uniform, init=RangeEstimators.current_minmax, qparams=dict(n_bits=8))
set_act_quant_axis_and_groups(mgr, axis=1, n_groups=None)
with torch.no_grad():
    for _ in range(50): mgr(x)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(2000): mgr(x)
    torch.cuda.synchronize(); print('dynamic per-token manager call [8,128,768] fp32: %.1f us' % ((time.perf_counter()-t0)/2000*1e6))
PY
