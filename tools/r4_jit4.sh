#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for g in 0 1; do FG_MH_GEN_ALL=$g AB_PIPES=0 timeout -k 10 300 python tools/ab_mh_pipe.py c5 262144 32768 2>&1 | grep -v amdgpu.ids | sed "s/^/gen_all=$g /"; done | tee gpurun_out/r4_c5_gen.txt
for gm in 8 4 2; do FG_MH_GEN_MIN=$gm timeout -k 10 600 python tools/bench_jit_all.py refmodel8 2>&1 | grep -v amdgpu.ids | sed "s/^/gen_min=$gm /"; done | tee -a gpurun_out/r4_c5_gen.txt
