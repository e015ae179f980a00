#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 500 python tools/ab_smc_zoom.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_smc_zoom.txt
