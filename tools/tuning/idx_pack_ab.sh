R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/idx_pack_ab.txt; : > $O
for rep in 1 2; do for which in shipped old; do
  if [ $which = old ]; then export TQ_LIB_PATH=$R/tools/tuning/_ab/libtq_hip.so; else unset TQ_LIB_PATH; fi
  echo "== $which (rep $rep)" >> $O
  python $R/tools/tuning/idx_only_time.py 2>&1 | grep -v amdgpu.ids >> $O
done; done
unset TQ_LIB_PATH
python -m pytest tests/test_hip_parity.py tests/test_per_token.py tests/test_linear_i8.py -m gpu -x -q 2>&1 | tail -4 >> $O
cat $O
