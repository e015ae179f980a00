#!/bin/bash
# phase cycle counters of the MH and HMC multi-wave kernels (experiment build libfugue_prof.so, built beforehand:
#   FG_LIB_PATH=$PWD/fugue_amd/lib/libfugue_prof.so FG_EXTRA_DEFS=FG_MH_PROF,FG_HMC_PROF python -m fugue_amd.build)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export FG_LIB_PATH=$PWD/fugue_amd/lib/libfugue_prof.so FG_EXTRA_DEFS=FG_MH_PROF,FG_HMC_PROF
( for c in 65536 8192; do timeout -k 10 200 python tools/prof_mh_phases.py ref $c 2>&1 | grep -v amdgpu.ids; done
  timeout -k 10 200 python tools/prof_mh_phases.py c5 2>&1 | grep -v amdgpu.ids ) | tee gpurun_out/r4_mh_phases_after.txt
( for c in 65536 8192; do timeout -k 10 200 python tools/prof_hmc_phases.py $c 2>&1 | grep -v amdgpu.ids; done ) | tee gpurun_out/r4_hmc_phases_after.txt
