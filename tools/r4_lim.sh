#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "larger_than_one_lds_tile" --durations=3 2>&1 | tail -12
timeout -k 10 900 python -m pytest tests/test_gpu_smc.py -x -q -k multisite 2>&1 | tail -5
