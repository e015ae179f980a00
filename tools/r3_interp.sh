#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3_interp_test.log 2>&1; echo "test rc $?"
tail -8 gpurun_out/r3_interp_test.log
