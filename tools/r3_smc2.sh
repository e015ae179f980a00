#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_smc.py -x -q > gpurun_out/r3_smc_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3_smc_tests.log
tail -4 gpurun_out/r3_smc_tests.log
grep -q "tests rc 0" gpurun_out/r3_smc_tests.log || exit 1
python tools/bench_smc.py 2>&1 | grep "smc " | tee gpurun_out/r3_smc_bench.txt
bash tools/prof_smc_phases.sh | sed -n 4,8p
