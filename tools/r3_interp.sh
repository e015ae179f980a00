#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
ulimit -c 0; export HSA_ENABLE_COREDUMP=0
timeout -k 10 900 python -m pytest tests/test_gpu_jit.py -x -q -m gpu > gpurun_out/r3_interp_test.log 2>&1; echo "test rc $?"
tail -5 gpurun_out/r3_interp_test.log
timeout -k 10 900 python tools/bench_jit_big.py 100 1000 10000 30000 2>&1 | tee gpurun_out/r3_jit_big.log
