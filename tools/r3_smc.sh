#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_smc.py tests/test_gpu_diag.py -x -q > gpurun_out/r3_smc_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3_smc_tests.log
tail -4 gpurun_out/r3_smc_tests.log
grep -q "tests rc 0" gpurun_out/r3_smc_tests.log || exit 1
python tools/bench_smc.py 2>&1 | grep "smc " | tee gpurun_out/r3_smc_bench.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_smc_prof -- python3 $R/tools/bench_smc.py > $R/gpurun_out/r3_smc_prof.log 2>&1
cd $R && python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3_smc_prof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("kernel time per run %.1f us over %.0f launches" % (tot / 4 / 1e3, sum(int(r["Calls"]) for r in rows) / 4))
for r in rows[:14]:
    print("%-50s calls/run %5.1f  avg %7.2f us  total/run %7.1f us" % (r["Name"][:50], int(r["Calls"]) / 4, float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 4 / 1e3))
PY
