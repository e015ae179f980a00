#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "test_hmc_sep_kernel_is_bit_identical" 2>&1 | tail -6 &&
(echo "FG_HMC_DENSE_FAST=0"; FG_HMC_DENSE_FAST=0 timeout -k 10 300 python tools/bench_dense.py 2>&1 | grep -v amdgpu.ids; echo "default"; timeout -k 10 300 python tools/bench_dense.py 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/r4_dense.txt
