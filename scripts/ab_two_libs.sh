#!/bin/bash
# Run ON THE GPU BOX: the shipped library against tools/tuning/_ab/libtq_hip.so (tools/tuning/build_ab_lib.py) on ONE box,
# alternating: default-route forwards and the latency-bound kernel families.  -> gpurun_out/ab_two_libs.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_two_libs.txt
mkdir -p $R/gpurun_out; : > $O
for rep in 1 2 3; do
  for which in shipped ab; do
    if [ $which = ab ]; then export TQ_LIB_PATH=$R/tools/tuning/_ab/libtq_hip.so; else unset TQ_LIB_PATH; fi
    echo "== $which (rep $rep)" >> $O
    python $R/tools/tuning/fwd_ab.py 2>&1 | grep forward_ms >> $O
    if [ $rep = 1 ]; then python $R/scripts/kernel_bench.py --only i8,ada,dyn_small 2>&1 | grep -E "^(i8|ada|dyn)" >> $O; fi
  done
done
cat $O
