#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r4_jit_vs_stream_tasks_big.txt; : > $O
cat > /tmp/js.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from fugue_amd import engine as E, workloads as W
from tests.models import ZOO
names, sizes = sys.argv[1].split(","), [int(x) for x in sys.argv[2].split(",")]
for name in names:
    cp = E.compile_model(ZOO[name]() if name in ZOO else W.reference_model(int(name[8:])))
    for C in sizes:
        eng = E.Engine(cp, C, seed=2)
        eng.hmc_init(E.hmc_config(n_leapfrog=16), 10); eng.hmc_step(10); eng.synchronize()
        t0 = time.perf_counter(); eng.hmc_step(20); eng.synchronize(); dt = time.perf_counter() - t0
        print(f"FG_JIT={os.environ.get('FG_JIT')} SEP={os.environ.get('FG_HMC_SEP')} {name:12s} C={C:6d} {C * 20 * 16 / dt:.3e} [{eng.hmc_last_kernel()[:40]}]", flush=True)
        eng.close()
PY
for j in 0 2; do FG_JIT=$j timeout -k 10 400 python /tmp/js.py refmodel8,refmodel20,refmodel32 262144,524288 2>&1 | grep -v amdgpu.ids >> $O || exit 1; done
FG_JIT=1 timeout -k 10 400 python /tmp/js.py normal32 65536,8192 2>&1 | grep -v amdgpu.ids >> $O || exit 1
FG_JIT=2 FG_HMC_SEP=0 timeout -k 10 400 python /tmp/js.py normal32 65536,8192 2>&1 | grep -v amdgpu.ids >> $O || exit 1
cat $O
