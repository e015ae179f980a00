#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "interp_multiwave or test_hmc_chain or transition_injected" > gpurun_out/r3_interp_test.log 2>&1; echo "test rc $?"
tail -5 gpurun_out/r3_interp_test.log
timeout -k 10 120 python tools/mb_interp_costs.py 2 > gpurun_out/r3_interp_costs.log 2>&1; echo "costs rc $?"; cat gpurun_out/r3_interp_costs.log
timeout -k 10 600 python tools/bench_interp_mw.py "$@" > gpurun_out/r3_interp_bench.log 2>&1; echo "bench rc $?"
cat gpurun_out/r3_interp_bench.log
