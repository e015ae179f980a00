#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_jit.py tests/test_gpu_mh.py -x -q -m gpu > gpurun_out/r3_interp_test.log 2>&1; echo "test rc $?"
tail -15 gpurun_out/r3_interp_test.log
timeout -k 10 600 python tools/bench_mh_interp.py > gpurun_out/r3_mh_interp_bench.log 2>&1; echo "bench rc $?"
cat gpurun_out/r3_mh_interp_bench.log
