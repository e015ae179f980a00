#!/bin/bash
# round 3: full GPU suite, the two-rank rehearsal, the default bench line
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
ulimit -c 0; export HSA_ENABLE_COREDUMP=0      # a faulting kernel must not write a core dump of the GPU (tens of GB)
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_suite.log 2>&1; echo "suite rc $?" | tee -a gpurun_out/r3_gpu_suite.log
tail -4 gpurun_out/r3_gpu_suite.log
grep -q "suite rc 0" gpurun_out/r3_gpu_suite.log || exit 1
timeout -k 10 600 bash tools/rehearse_ranks.sh > gpurun_out/r3_rehearse.log 2>&1; echo "rehearse rc $?" | tee -a gpurun_out/r3_rehearse.log
tail -6 gpurun_out/r3_rehearse.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 900 python bench.py > gpurun_out/r3_bench_a.json 2> gpurun_out/r3_bench_a.err; echo "bench rc $?"
tail -12 gpurun_out/r3_bench_a.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r3_bench_a.json").read().strip().splitlines()[-1])
print("value", j["value"], "frac", j["roofline"]["frac"], j["roofline"]["kernel"], j["timed_regions"]["value"])
for k in ("c3", "hmc_fd_dense", "mh", "c5", "smc"):
    v = j.get(k, {})
    if k == "c3":
        print(k, {t: (v[t]["value"], v[t]["roofline"]["frac"], v[t]["roofline"]["kernel"]) for t in ("chains_65536", "chains_8192")}, v.get("cpu_baseline", {}).get("value"))
    else:
        print(k, v.get("value"), v.get("roofline", {}).get("frac"), v.get("cpu_baseline", {}).get("value"), v.get("vs_cpu_baseline"))
print("cpu", j.get("cpu_baseline", {}).get("value"), "validity", j.get("validity"))
PY
