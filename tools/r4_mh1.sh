#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_mh.py -x -q > gpurun_out/r4_mh1_tests.txt 2>&1; echo "tests rc=$?" ; tail -15 gpurun_out/r4_mh1_tests.txt
timeout -k 10 300 python tools/ab_mh_pipe.py all 65536 16384 8192 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_mh1_ab.txt
