#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 300 python tools/ab_smc_zoom.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_smc_zoom.txt &&
timeout -k 10 900 python -m pytest tests/test_gpu_smc.py tests/test_gpu_fullsize.py -x -q -k "smc or larger" 2>&1 | tail -8 &&
timeout -k 10 300 python tools/bench_smc.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_smc_fused2.txt
