#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3_interp_test.log 2>&1; echo "test rc $?"
tail -5 gpurun_out/r3_interp_test.log
timeout -k 10 600 python tools/bench_interp_mw.py > gpurun_out/r3_interp_bench.log 2>&1; echo "bench rc $?"
cat gpurun_out/r3_interp_bench.log
timeout -k 10 600 python tools/bench_mh_interp.py > gpurun_out/r3_mh_interp_bench.log 2>&1; echo "bench rc $?"
grep -v "multi-wave=[248]" gpurun_out/r3_mh_interp_bench.log
