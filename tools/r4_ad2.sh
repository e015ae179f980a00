#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_jit.py -x -q 2>&1 | tail -4 &&
timeout -k 10 600 python tools/bench_analytic.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_analytic2.txt
