#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 &&
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --full-out gpurun_out/r4_bench_c_full.json > gpurun_out/r4_bench_c.json 2> gpurun_out/r4_bench_c.err && wc -c gpurun_out/r4_bench_c.json && python - <<'PY'
import json
d=json.load(open('gpurun_out/r4_bench_c.json'))
print(d['value'], d['roofline']['frac'], d['cpu_baseline']['value'])
for k,v in d['legs'].items(): print(k, v)
PY
