#!/bin/bash
# Run ON THE GPU BOX (via gpurun): SQ counters of the kernels of one scripts/kernel_bench.py family, separate --pmc
# passes with --kernel-trace only (never combined with the runtime / memory-copy traces) -> gpurun_out/<tag>_pmc.json:
# per kernel name (first 90 characters) the mean counter values over its dispatches.
#   bash scripts/pmc_kernels.sh <family> <tag>
FAM=${1:-tails}
TAG=${2:-r03_$FAM}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${TAG}_pmc.json
rm -f /tmp/pmc_acc.json
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  rm -rf /tmp/pmc
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc -o p --output-format csv -- python $R/scripts/kernel_bench.py --only $FAM > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob, collections, json, os
acc = json.load(open('/tmp/pmc_acc.json')) if os.path.exists('/tmp/pmc_acc.json') else {}
tmp = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('/tmp/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'][:90]
        if k.startswith(('void at::', 'at::', 'void (anonymous')) or 'elementwise' in k or 'distribution' in k:
            continue
        tmp[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in tmp.items():
    for c, v in d.items():
        acc.setdefault(k, {})[c] = round(sum(v) / len(v), 1)
        acc[k]['_dispatches'] = len(v)
json.dump(acc, open('/tmp/pmc_acc.json', 'w'), indent=1, sort_keys=True)
PY
done
cp /tmp/pmc_acc.json $OUT
python3 - "$OUT" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, c in d.items():
    wc, va, vi = c.get('SQ_WAVE_CYCLES', 0), c.get('SQ_ACTIVE_INST_VALU', 0), c.get('SQ_INSTS_VALU', 0)
    print(f"{k[:70]:70s} waves {c.get('SQ_WAVES', 0):9.0f} valu_insts {vi:12.0f} valu_active/wave_cycles {va / wc if wc else 0:5.2f} "
          f"wait_any/wave_cycles {c.get('SQ_WAIT_ANY', 0) / wc if wc else 0:5.2f} busy {c.get('SQ_BUSY_CYCLES', 0):10.0f} gui {c.get('GRBM_GUI_ACTIVE', 0):9.0f}")
PY
