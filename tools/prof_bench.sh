#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command (what BENCH_rNN.json is measured with, minus the CPU baseline and extras)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.log
cd $R && head -4 gpurun_out/prof_bench/*/*kernel_stats.csv | cut -c1-220 && python3 -c "
import json
j = json.loads(open('gpurun_out/prof_bench.json').read().strip().splitlines()[-1]); print('bench under the profiler: avg_launch_ms', j['roofline']['avg_launch_ms'], 'value', j['value'])"
