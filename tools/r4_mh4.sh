#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export FG_LIB_PATH=$PWD/fugue_amd/lib/libfugue_prof.so FG_EXTRA_DEFS=FG_MH_PROF,FG_HMC_PROF FG_JIT=0
for c in 8192; do python tools/prof_mh_phases.py ref $c 2>&1 | grep -v amdgpu.ids | grep -v "wave  [4-9]\|wave 1[0-3]"; done | tee gpurun_out/r4_mh_phases_pipe1c.txt
