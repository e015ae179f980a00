#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for sm in 0 1; do FG_MH_JIT_SUMS=$sm AB_PIPES=0 timeout -k 10 300 python tools/ab_mh_pipe.py ref 65536 8192 2>&1 | grep -v amdgpu.ids | sed "s/^/jit_sums=$sm /"; done | tee gpurun_out/r4_mh_jitsums2.txt
