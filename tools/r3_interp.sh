#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_jit.py -x -q -m gpu > gpurun_out/r3_interp_test.log 2>&1; echo "test rc $?"
tail -5 gpurun_out/r3_interp_test.log
timeout -k 10 900 python tools/bench_jit_big.py 1000 10000 2>&1 | tee gpurun_out/r3_jit_big.log
