#!/bin/bash
# k_hmc_jit_steps: whole coordinates per wave, one barrier per gradient (FG_JIT_FUSED=1) against the task code with two barriers (FG_JIT_FUSED=0)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_jit.py -x -q 2>&1 | tail -3 || exit 1
O=gpurun_out/r4_hmc_jit_fused.txt; : > $O
for cfg in "X=default"; do
  echo "---- $cfg" >> $O
  env $cfg timeout -k 10 500 python tools/bench_jit_all.py alldists logistic poisson_glm hier_logsigma hier_scale linreg mixture refmodel8 hier refmodel20 refmodel32 2>&1 | grep -v amdgpu.ids | sed 's/  MH .*//' >> $O || exit 1
done
cat $O
