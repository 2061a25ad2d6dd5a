#!/bin/bash
# Run ON THE GPU BOX (via gpurun): everything profiles/r06 is made of, from ONE box.  ~8 minutes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd "$ROOT" || exit 1
O=$ROOT/gpurun_out/r06
mkdir -p "$O"
export TMPDIR=/tmp
python tools/tuning/py_overhead2.py 2>&1 | grep -v amdgpu > "$O/py_overhead2.txt"
# the PMC passes first: bench.py reads profiles/pmc_traffic.json and refuses one collected from other kernel sources
bash scripts/profile_gpu.sh r06 > "$O/profile_gpu.log" 2>&1
cp gpurun_out/prof_r06/pmc_traffic.json profiles/pmc_traffic.json
python bench.py > "$O/bench_n1.json" 2> "$O/bench_n1.err"; echo "bench rc=$?"
python bench.py --sweep --no-cpu --headline-only > "$O/bench_sweep.json" 2>> "$O/bench_n1.err"
TQ_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 \
    bench.py --gpus 1 --no-cpu > "$O/bench_rccl1.json" 2> "$O/bench_rccl1.err"; echo "bench rccl1 rc=$?"
bash scripts/profile_kernels.sh r06 > "$O/profile_kernels.log" 2>&1
python scripts/config_bench.py > "$O/config_bench.json" 2> "$O/config_bench.err"; echo "config_bench rc=$?"
python scripts/int_vs_reference.py > "$O/int_vs_reference.json" 2> "$O/int_vs_reference.err"; echo "int_vs_reference rc=$?"
rm -rf /tmp/layer_tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/layer_tl -o t -- python tools/tuning/bert_default_prof.py > /dev/null 2>&1
python tools/tuning/layer_timeline.py /tmp/layer_tl > "$O/bert_default_route_layer_timeline.txt"
rm -rf /tmp/layer_tl_mb; rocprofv3 --kernel-trace --output-format csv -d /tmp/layer_tl_mb -o t -- python tools/tuning/mb_default_prof.py > /dev/null 2>&1
python tools/tuning/layer_timeline.py /tmp/layer_tl_mb mobilebert > "$O/mobilebert_default_route_layer_timeline.txt"
timeout 1500 python -m pytest tests -q -m gpu -rs > "$O/gpu_tests_full_suite.log" 2>&1; echo "suite rc=$?"
tail -3 "$O/gpu_tests_full_suite.log"
head -c 1200 "$O/bench_n1.json"; echo
tail -3 "$O/config_bench.err"
cat "$O/py_overhead2.txt"
# round 6: the attention core's phase profile (instrumented build next to the tool), graph-replay timings and counters
python tools/tuning/attn_prof.py 2>&1 | grep -v amdgpu.ids > "$O/attn_phase_profile.txt"
python tools/tuning/attn_graph.py 2>&1 | grep -v amdgpu.ids > "$O/attn_graph.txt"
bash scripts/pmc_attention.sh 64 > "$O/pmc64.log" 2>&1; cp gpurun_out/attention_pmc_B64.json "$O/attention_pmc_B64.json"
bash scripts/pmc_attention.sh 8 > "$O/pmc8.log" 2>&1; cp gpurun_out/attention_pmc_B8.json "$O/attention_pmc_B8.json"
cat "$O/attn_graph.txt"
