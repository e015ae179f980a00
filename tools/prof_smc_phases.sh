#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
L=$PWD/gpurun_out/exp_lib_smcprof.so
FG_LIB_PATH=$L FG_EXTRA_DEFS=FG_SMC_PROF python -c "from fugue_amd import build; build.build()" || exit 1
FG_LIB_PATH=$L python tools/prof_smc_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_smc_phases.txt
