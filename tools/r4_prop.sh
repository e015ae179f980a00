#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 300 python tools/bench_jit_all.py alldists logistic 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_prop.txt &&
timeout -k 10 900 python -m pytest tests/test_gpu_jit.py tests/test_gpu_mh.py -x -q 2>&1 | tail -4
