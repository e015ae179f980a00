#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 500 python tools/fuzz_hmc_sep.py 11 70 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_fuzz_hmc_sep.txt; tail -3 gpurun_out/r4_fuzz_hmc_sep.txt; grep -c "registers" gpurun_out/r4_fuzz_hmc_sep.txt
timeout -k 10 300 python tools/fuzz_mh_mw.py 5 30 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_fuzz_mh_mw.txt; tail -2 gpurun_out/r4_fuzz_mh_mw.txt
