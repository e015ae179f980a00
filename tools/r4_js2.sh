#!/bin/bash
# stream kernels (FG_JIT=0 at these sizes where the rule prefers the unit) against the compiled unit (FG_JIT=2) after the straight-line task lists
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_jit.py -x -q 2>&1 | tail -2 || exit 1
O=gpurun_out/r4_jit_vs_stream_tasks.txt; : > $O
cat > /tmp/js.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from fugue_amd import engine as E, workloads as W
from tests.models import ZOO
for name in ("refmodel8", "refmodel20", "refmodel32", "hier_scale", "mixture", "linreg"):
    cp = E.compile_model(ZOO[name]() if name in ZOO else W.reference_model(int(name[8:])))
    for C in (131072, 65536, 32768):
        eng = E.Engine(cp, C, seed=2)
        eng.hmc_init(E.hmc_config(n_leapfrog=16), 10); eng.hmc_step(10); eng.synchronize()
        t0 = time.perf_counter(); eng.hmc_step(20); eng.synchronize(); dt = time.perf_counter() - t0
        print(f"FG_JIT={os.environ.get('FG_JIT')} {name:12s} C={C:6d} {C * 20 * 16 / dt:.3e} [{eng.hmc_last_kernel()[:40]}]", flush=True)
        eng.close()
PY
for j in 0 2; do FG_JIT=$j timeout -k 10 400 python /tmp/js.py 2>&1 | grep -v amdgpu.ids >> $O || exit 1; done
sort -k2,2 -k3,3 -s $O
