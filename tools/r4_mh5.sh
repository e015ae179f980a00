#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 600 python tools/ab_mh_pipe.py all 4096 8192 16384 32768 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_mh5_ab.txt
