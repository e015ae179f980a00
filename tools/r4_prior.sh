#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_jit.py tests/test_gpu_smc.py tests/test_gpu_mh.py tests/test_gpu_session.py tests/test_gpu_state.py -x -q 2>&1 | tail -5 &&
timeout -k 10 300 python tools/bench_smc.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_smc_prior.txt &&
timeout -k 10 300 python tools/bench_smc_interp.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4_smc_prior.txt
