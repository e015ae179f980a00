#!/bin/bash
# round 3: the run-time compiled kernels -- their tests, the comparison with the interpreter / stream kernels, large programs
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
ulimit -c 0; export HSA_ENABLE_COREDUMP=0      # a faulting kernel must not write a core dump of the GPU (tens of GB)
timeout -k 10 900 python -m pytest tests/test_gpu_jit.py tests/test_gpu_mh.py -x -q -m gpu > gpurun_out/r3_jit_test.log 2>&1; echo "test rc $?"
tail -4 gpurun_out/r3_jit_test.log
timeout -k 10 600 python tools/bench_interp_mw.py 2>&1 | tee gpurun_out/r3_interp_bench.log | tail -50
timeout -k 10 600 python tools/bench_mh_interp.py 2>&1 | tee gpurun_out/r3_mh_interp_bench.log | tail -40
timeout -k 10 600 python tools/bench_jit_vs_stream.py 2>&1 | tee gpurun_out/r3_jit_vs_stream.log | tail -40
timeout -k 10 900 python tools/bench_jit_big.py 100 1000 10000 2>&1 | tee gpurun_out/r3_jit_big.log
