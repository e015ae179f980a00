#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_smc_pmc; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $O/p1 -- python3 $R/tools/bench_smc.py > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python3 $R/tools/bench_smc.py > $O/p2.log 2>&1
cd $R && python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/r3_smc_pmc/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "ess2" in k or "rejuv" in k or "red_sum" in k:
            print(p, k, {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())}, "n", len(next(iter(cs.values()))))
PY
