#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_jit.py -x -q -k "dense_mode or stream_kernels" 2>&1 | tail -3 || exit 1
O=gpurun_out/r4_hmc_jit_dense_one_barrier.txt; : > $O
cat > /tmp/jd.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from fugue_amd import engine as E, workloads as W
from tests.models import ZOO
for name in ("refmodel8", "hier_scale", "mixture", "poisson_glm", "logistic", "refmodel32", "hier"):
    cp = E.compile_model(ZOO[name]() if name in ZOO else W.reference_model(int(name[8:])))
    for C in (65536, 8192):
        eng = E.Engine(cp, C, seed=2)
        eng.hmc_init(E.hmc_config(n_leapfrog=16, grad_mode=E.GRAD_FD_DENSE), 5); eng.hmc_step(5); eng.synchronize()
        t0 = time.perf_counter(); eng.hmc_step(10); eng.synchronize(); dt = time.perf_counter() - t0
        print(f"FUSED={os.environ.get('FG_JIT_FUSED', 'rule'):4s} {name:12s} C={C:6d} {C * 10 * 16 / dt:.3e} [{eng.hmc_last_kernel()[16:80]}]", flush=True)
        eng.close()
PY
for f in 0 1 rule; do if [ $f = rule ]; then timeout -k 10 400 python /tmp/jd.py 2>&1 | grep -v amdgpu.ids >> $O || exit 1; else FG_JIT_FUSED=$f timeout -k 10 400 python /tmp/jd.py 2>&1 | grep -v amdgpu.ids >> $O || exit 1; fi; done
sort -k2,2 -k3,3 -s $O
