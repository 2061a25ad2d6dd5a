#!/bin/bash
# Run HERE after `gpurun -- bash scripts/rNN_collect.sh`: copy what the collection wrote under gpurun_out/ into the
# tracked profiles/rNN (and profiles/pmc_traffic.json, the profile bench.py reads), then regenerate the README.
#   bash scripts/install_profiles.sh r05
set -eu
R=${1:?round tag, e.g. r05}
cd "$(dirname "$0")/.."
G=gpurun_out
P=profiles/$R
mkdir -p "$P"
for f in bench_n1.json bench_rccl1.json bench_sweep.json config_bench.json int_vs_reference.json gpu_tests_full_suite.log \
         py_overhead2.txt mx_probe.txt bert_default_route_layer_timeline.txt mobilebert_default_route_layer_timeline.txt \
         attn_phase_profile.txt attn_graph.txt attention_pmc_B64.json attention_pmc_B8.json; do
  [ -s "$G/$R/$f" ] && cp "$G/$R/$f" "$P/$f"
done
cp "$G/prof_$R/summary.txt" "$P/summary.txt"
cp "$G/prof_$R/summary_$R.json" "$P/summary_$R.json"
cp "$G/prof_$R/pmc_traffic.json" "$P/pmc_traffic.json"
cp "$G/prof_$R/pmc_traffic.json" profiles/pmc_traffic.json
cp "$G/prof_$R/trace/"*kernel_stats.csv "$P/kernel_stats.csv"
cp "$G/prof_kernels_$R/table/kernel_table.md" "$P/kernel_table.md"
cp "$G/prof_kernels_$R/table/kernel_table.json" "$P/kernel_table.json"
cp "$G/prof_kernels_$R/manifest.json" "$P/kernel_bench_manifest.json"
cp "$G/prof_kernels_$R/trace/"*kernel_stats.csv "$P/kernel_bench_kernel_stats.csv"
python scripts/summarize_profiles.py readme "$P"
git status --short profiles | head -40
