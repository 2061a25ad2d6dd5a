#!/bin/bash
# Exercise validate_quantized.py over the reference's flag combinations on a 1-layer model (run through gpurun).
B="python transformer-quantization_amd/validate_quantized.py --qmethod symmetric_uniform --qmethod-act asymmetric_uniform --num-layers 1 --num-eval-batches 1"
run() { echo "== $*"; timeout 300 $B "$@" 2>&1 | grep -v "amdgpu\|^INFO" | tail -1 | python -c "
import json,sys
t=sys.stdin.read().strip()
try:
    d=json.loads(t); print('   OK', d['fidelity_vs_fp32'], d['timings_s'])
except Exception as e:
    print('   FAILED:', t[-300:])"; }
run --act-quant-method MSE --act-num-candidates 50 --num-est-batches 2
run --act-quant-method MSE --act-opt-method golden_section
run --act-quant-method current_minmax --percentile 0.01    # reference quirk q6: (p, 100) percentiles, so p is the LOWER tail
run --weight-quant-method MSE --per-channel --num-candidates 40
run --weight-quant-method MSE --weight-opt-method golden_section
run --cross-entropy-layer classifier --num-est-batches 2
run --per-token
run --per-embd --act-quant-method current_minmax
run --per-groups 6 --per-groups-permute-shared-h --act-quant-method current_minmax   # phase-1 ranges exist only in CurrentMinMaxEstimator (as upstream)
run --no-act-quant
run --no-weight-quant
run --dynamic
run --qmethod asymmetric_uniform --n-bits 6
run --n-bits 4 --adaround layers.0.output.dense --adaround-iters 20 --adaround-num-samples 16 --adaround-mode learned_sigmoid --adaround-init mse
run --n-bits 4 --adaround layers.0.output.dense --adaround-iters 20 --adaround-num-samples 16 --adaround-mode sigmoid_temp_decay --adaround-init mse_out --adaround-no-act-func
run --n-bits 4 --adaround layers.0.intermediate.0 --adaround-iters 20 --adaround-num-samples 16 --no-adaround-asym --adaround-act-quant no_act_quant
run --act-quant-method current_minmax --quant-dict "{'y': 'ngp6', 'h': 16, 'Et': 4, 's': 'fp32', 'wC': 'fp32', 'x0': 'per_embd'}"
run --fast-inference --hip-graph --num-est-batches 3
run --double --num-est-batches 2
run --double --per-embd --act-quant-method current_minmax --weight-quant-method MSE
run --double --act-quant-method MSE --act-num-candidates 20
