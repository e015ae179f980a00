#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3_interp_test.log 2>&1; echo "test rc $?"
tail -6 gpurun_out/r3_interp_test.log
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from fugue_amd import engine as E, workloads as W
from tests.models import ZOO
for name, prog in [("alldists", ZOO["alldists"]()), ("logistic100", W.logistic_regression(*W.classification_data(100)[:2])), ("logistic1000", W.logistic_regression(*W.classification_data(1000)[:2]))]:
    cp = E.compile_model(prog)
    for jit in (0, 1):
        os.environ["FG_JIT"] = str(jit)
        eng = E.Engine(cp, 65536, seed=1)
        eng.hmc_init(E.hmc_config(), 2); eng.synchronize()          # compile + first init
        t0 = time.perf_counter(); eng.hmc_init(E.hmc_config(), 2); eng.synchronize(); dt = time.perf_counter() - t0
        print(f"{name:12s} FG_JIT={jit}  fg_hmc_init (step-size search) {dt*1e3:8.1f} ms", flush=True)
        eng.close()
PY
