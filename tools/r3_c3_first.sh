#!/bin/bash
# round 3: first run of the observation-major regression kernel -- parity, then C3 throughput old vs new
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "lin_kernel or ridge" > gpurun_out/r3_lin_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3_lin_tests.log
tail -5 gpurun_out/r3_lin_tests.log
for lin in 0 1; do
  for ch in 65536 8192; do
    FG_HMC_LIN=$lin timeout -k 10 300 python tools/bench_c3.py --chains $ch --transitions 2 $( [ $ch = 65536 ] && [ $lin = 1 ] && echo --check ) 2>&1 | sed "s/^/LIN=$lin /" | tee -a gpurun_out/r3_c3_first.txt
  done
done
