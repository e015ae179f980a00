#!/bin/bash
# HBM traffic of the bench kernels: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (TCC slots), kernel-trace only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python3 $R/bench.py --steps 100 --warmup 50 --launch 25 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- $CMD > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- $CMD > $R/gpurun_out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof2 -- $CMD > $R/gpurun_out/prof2.log 2>&1
cd $R && python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_fetch", "gpurun_out/pmc_write"):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        if "k_" in k[0]:
            print(d, k, "mean=%.1f" % (sum(v) / len(v)), "n=%d" % len(v), "first=%.1f last=%.1f" % (v[0], v[-1]))
PY
cat gpurun_out/prof2/*/*kernel_stats.csv | head -4 | cut -c1-160
