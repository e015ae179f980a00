#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 300 python tools/bench_smc_interp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_smc_prior2.txt
timeout -k 10 100 python tools/bench_smc.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4_smc_prior2.txt
bash tools/prof_round.sh b2 "smc|c4|1048576" | tail -22
