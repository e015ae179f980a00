#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_jit.py tests/test_gpu_mh.py -x -q 2>&1 | tail -3
timeout -k 10 600 python tools/ab_mh_gen.py 2>&1 | grep -v amdgpu.ids | grep "gen_all=1\|c5" | tee gpurun_out/r4_mh_gen_matrix3.txt
