#!/bin/bash
# round-4 first GPU visit: new tests, the new bench line (plain + under a forced 1-rank RCCL group)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_rccl_raw.py tests/test_rccl_single_rank.py tests/test_qat_step.py -m gpu -x -q > gpurun_out/r04/tests_new.log 2>&1
echo "tests rc=$?" >> gpurun_out/r04/tests_new.log
tail -15 gpurun_out/r04/tests_new.log
timeout 600 python bench.py > gpurun_out/r04/bench_n1.json 2> gpurun_out/r04/bench_n1.err
echo "bench rc=$?"; tail -c 3000 gpurun_out/r04/bench_n1.json; tail -5 gpurun_out/r04/bench_n1.err
TQ_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --no-cpu > gpurun_out/r04/bench_rccl1.json 2> gpurun_out/r04/bench_rccl1.err
echo "bench rccl1 rc=$?"; tail -c 3000 gpurun_out/r04/bench_rccl1.json; tail -5 gpurun_out/r04/bench_rccl1.err
