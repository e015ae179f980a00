#!/bin/bash
# the lin kernel's waves-per-tile layouts side by side: eight coordinates per wave (D / 8 waves) against four (D / 4)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "lin_kernel" 2>&1 | tail -4 || exit 1
O=gpurun_out/r4_lin_eight.txt; : > $O
run() { echo "p=$1 chains $2 FG_HMC_WAVES=$3" >> $O; FG_HMC_WAVES=$3 timeout -k 10 300 python tools/bench_c3.py --p $1 --chains $2 --transitions $4 2>&1 | grep -v amdgpu.ids >> $O || exit 1; }
for ch in 65536 8192 4096; do run 32 $ch 8 4; run 32 $ch 4 4; done
for ch in 65536 8192; do run 16 $ch 4 6; run 16 $ch 2 6; done
run 64 65536 16 1; run 64 65536 8 1
run 64 16384 16 2; run 64 16384 8 2
run 40 65536 16 1; run 40 65536 8 1
cat $O
