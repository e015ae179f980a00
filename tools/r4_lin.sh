#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "lin_kernel" 2>&1 | tail -8
for p in 32 64 24; do timeout -k 10 300 python tools/bench_c3.py --p $p --transitions 3 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r4_lin_bench.txt
timeout -k 10 300 python tools/bench_c3.py --p 32 --chains 8192 --transitions 6 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4_lin_bench.txt
FG_HMC_LIN=0 timeout -k 10 300 python tools/bench_c3.py --p 64 --chains 16384 --transitions 1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4_lin_bench.txt
