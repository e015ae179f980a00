#!/bin/bash
# waves per tile of k_hmc_jit_steps after the task-code change (FG_HMC_INTERP_WAVES overrides the host rule)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r4_hmc_jit_waves.txt; : > $O
cat > /tmp/jw.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from fugue_amd import engine as E, workloads as W
from tests.models import ZOO
for name in ("refmodel8", "refmodel20", "refmodel32", "hier", "linreg", "hier_scale", "mixture"):
    cp = E.compile_model(ZOO[name]() if name in ZOO else W.reference_model(int(name[8:])))
    for C in (65536, 8192):
        eng = E.Engine(cp, C, seed=2)
        eng.hmc_init(E.hmc_config(n_leapfrog=16), 10); eng.hmc_step(10); eng.synchronize()
        t0 = time.perf_counter(); eng.hmc_step(20); eng.synchronize(); dt = time.perf_counter() - t0
        print(f"W={os.environ.get('FG_HMC_INTERP_WAVES', 'rule'):4s} {name:12s} C={C:6d} {C * 20 * 16 / dt:.3e} [{eng.hmc_last_kernel()[:22]}]", flush=True)
        eng.close()
PY
for w in rule; do if [ $w = rule ]; then timeout -k 10 400 python /tmp/jw.py 2>&1 | grep -v amdgpu.ids >> $O || exit 1; else FG_HMC_INTERP_WAVES=$w timeout -k 10 400 python /tmp/jw.py 2>&1 | grep -v amdgpu.ids >> $O || exit 1; fi; done
sort -k2,2 -k3,3 -s $O
