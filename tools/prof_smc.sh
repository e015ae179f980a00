#!/bin/bash
# rocprofv3 --kernel-trace --stats of adaptive_smc at 1 048 576 particles (tools/bench_smc.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_smc
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_smc -- python3 $R/tools/bench_smc.py > $R/gpurun_out/prof_smc.log 2>&1
cd $R && cat gpurun_out/prof_smc.log | grep -v amdgpu && cut -d, -f1-8 gpurun_out/prof_smc/*/*kernel_stats.csv | head -30
