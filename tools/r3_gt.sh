#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "larger_than or global_tile" > gpurun_out/r3_gt_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3_gt_tests.log
tail -30 gpurun_out/r3_gt_tests.log
