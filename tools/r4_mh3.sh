#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_mh.py -x -q -k "multiwave_kernel_is_identical or mixture_without" 2>&1 | tail -3
timeout -k 10 300 python tools/ab_mh_pipe.py ref 65536 8192 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_mh3_ab.txt
export FG_LIB_PATH=$PWD/fugue_amd/lib/libfugue_prof.so FG_EXTRA_DEFS=FG_MH_PROF,FG_HMC_PROF FG_JIT=0
for c in 65536 8192; do python tools/prof_mh_phases.py ref $c 2>&1 | grep -v amdgpu.ids | grep -v "wave  [4-9]\|wave 1[0-3]"; done | tee gpurun_out/r4_mh_phases_pipe1b.txt
