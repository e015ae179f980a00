#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "lin_kernel" > gpurun_out/r3_half_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3_half_tests.log
tail -4 gpurun_out/r3_half_tests.log
grep -q "tests rc 0" gpurun_out/r3_half_tests.log || exit 1
for ch in 4096 8192 12288 16384; do for half in 0 1; do for w in 8 16; do
  FG_HMC_LIN_HALF=$half FG_HMC_WAVES=$w timeout -k 10 300 python tools/bench_c3.py --chains $ch --transitions 3 2>&1 | sed "s/^/half=$half W=$w /" | tee -a gpurun_out/r3_half.txt
done; done; done
