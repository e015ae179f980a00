#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python bench.py --full-out gpurun_out/r4_bench_default_full.json > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err && wc -c gpurun_out/r4_bench_default.json &&
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --full-out gpurun_out/r4_bench_s20_full.json > gpurun_out/r4_bench_s20.json 2> gpurun_out/r4_bench_s20.err && wc -c gpurun_out/r4_bench_s20.json
