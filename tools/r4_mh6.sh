#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_mh.py -x -q 2>&1 | tail -3
AB_PIPES=0 timeout -k 10 600 python tools/ab_mh_pipe.py all 65536 8192 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_mh6_ab.txt
