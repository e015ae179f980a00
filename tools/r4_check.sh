#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_b.json 2> gpurun_out/r4_bench_b.err || { echo bench failed; tail -20 gpurun_out/r4_bench_b.err; exit 1; }
cp gpurun_out/bench_full.json gpurun_out/r4_bench_b_full.json
python - <<'PY'
import json
j = json.load(open("gpurun_out/r4_bench_b.json"))
print("headline %.4g" % j["value"], {k: "%.4g" % v["value"] for k, v in j["legs"].items()})
PY
