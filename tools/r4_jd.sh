#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_jit.py -x -q -k "dense_mode" 2>&1 | tail -6 &&
timeout -k 10 600 python tools/bench_jit_dense.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_jit_dense.txt
