#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for ns in 0 1; do FG_MH_NSEG_NS=$ns timeout -k 10 600 python tools/bench_jit_all.py alldists logistic poisson_glm hier_logsigma 2>&1 | grep -v amdgpu.ids | sed "s/^/nseg_ns=$ns /"; done | tee gpurun_out/r4_jit_ns.txt
for gm in 20 8; do FG_MH_GEN_MIN=$gm timeout -k 10 600 python tools/bench_jit_all.py refmodel8 2>&1 | grep -v amdgpu.ids | sed "s/^/gen_min=$gm /"; done | tee -a gpurun_out/r4_jit_ns.txt
