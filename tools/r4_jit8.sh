#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_jit.py tests/test_gpu_mh.py -x -q 2>&1 | tail -3
for g in 0 1; do FG_MH_GEN_ALL=$g AB_PIPES=0 timeout -k 10 300 python tools/ab_mh_pipe.py c5 262144 32768 2>&1 | grep -v amdgpu.ids | sed "s/^/gen_all=$g /"; done | tee gpurun_out/r4_c5_gen2.txt
timeout -k 10 300 python tools/bench_jit_all.py mixture alldists 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4_c5_gen2.txt
