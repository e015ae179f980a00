#!/bin/bash
# kernel-trace + --stats of the default bench command (CPU baselines skipped: they only add host time) -> gpurun_out/prof_<round>_bench
cd /tmp && export TMPDIR=/tmp
ROUND=${FG_PROF_ROUND:-round4}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_${ROUND}_bench; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --full-out $O/bench_full.json > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cd $R && python3 tools/prof_bench_region.py $O/trace $O/bench_full.json $O/${ROUND}_hmc_timed_region.txt
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/${ROUND}_bench_kernel_stats.csv
