#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 800 python tools/fuzz_jit.py 3 40 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_fuzz_jit.txt; tail -3 gpurun_out/r4_fuzz_jit.txt
