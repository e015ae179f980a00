#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sep_kernel_is_bit_identical" 2>&1 | tail -4
for s4 in 0 1; do for c in 4096 8192 16384; do FG_HMC_SUM4=$s4 FG_CHAINS=$c python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from fugue_amd import engine as E, workloads as W
C = int(os.environ["FG_CHAINS"])
cp = E.compile_model(W.normal_sites(32))
eng = E.Engine(cp, C, seed=1)
eng.hmc_init(E.hmc_config(), 100); eng.hmc_step(150); eng.synchronize()
best = 0
for _ in range(3):
    t0 = time.perf_counter(); eng.hmc_step(200); eng.synchronize(); best = max(best, C * 200 * 16 / (time.perf_counter() - t0))
print(f"sum4={os.environ['FG_HMC_SUM4']} C={C} {eng.hmc_last_kernel()}: {best:.4e} leapfrog-steps/s", flush=True)
PY
done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_sum4.txt
FG_LIB_PATH=$PWD/fugue_amd/lib/libfugue_prof.so FG_EXTRA_DEFS=FG_MH_PROF,FG_HMC_PROF python tools/prof_hmc_phases.py 8192 2>&1 | grep -v amdgpu.ids | grep "sampling\|wave  [0-4]:" | tail -6
