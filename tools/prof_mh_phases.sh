#!/bin/bash
# phase timing of the multi-wave MH kernel (experiment build with cycle counters; never the product library).
# Build first, where hipcc is:  FG_LIB_PATH=$PWD/fugue_amd/lib/libfugue_prof.so FG_EXTRA_DEFS=FG_MH_PROF,FG_HMC_PROF python -m fugue_amd.build
cd ${GRAFT_REPO_ROOT:-.}
export FG_LIB_PATH=$PWD/fugue_amd/lib/libfugue_prof.so FG_EXTRA_DEFS=FG_MH_PROF,FG_HMC_PROF
[ -f $FG_LIB_PATH ] || python -m fugue_amd.build > /dev/null 2>&1 || { echo build failed; exit 1; }
for m in ref c5; do python tools/prof_mh_phases.py $m 2>&1 | grep -v amdgpu.ids; done
python tools/prof_hmc_phases.py 2>&1 | grep -v amdgpu.ids
