#!/bin/bash
# Run ON THE GPU BOX (via gpurun): SQ counters of the integer attention core (T = 128, 12 heads of 64) at batch $1
# (default 64), separate --pmc passes with --kernel-trace only -> gpurun_out/attention_pmc_B$1.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-64}
OUT=$R/gpurun_out/attention_pmc_B$B.json
echo "{" > $OUT
first=1
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_WAVES"; do
  rm -rf /tmp/pmc
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc -o p --output-format csv -- python $R/tools/tuning/attn_one.py $B > /dev/null 2>&1
  python3 - "$OUT" "$first" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for fn in glob.glob('/tmp/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'attention_i8' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
with open(sys.argv[1], 'a') as f:
    for i, (k, v) in enumerate(sorted(acc.items())):
        f.write(('' if (sys.argv[2] == '1' and i == 0) else ',\n') + f' "{k}": {sum(v) / len(v):.1f}')
PY
  first=0
done
cat >> $OUT <<EOF2
,
 "_kernel": "tq::attention_i8_k, B=$B T=128 H=12 d=64, score / probability / context quantizers on, zero mask; average over 10 launches",
 "_units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles summed over all waves; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs; SQ_INSTS_* wave-instructions"
}
EOF2
cat $OUT
