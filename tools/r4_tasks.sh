#!/bin/bash
# the compiled HMC unit with its task split as straight-line code per wave (default) against the task list in memory (FG_JIT_TASKS=0)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_jit.py -x -q 2>&1 | tail -3 || exit 1
O=gpurun_out/r4_hmc_jit_tasks.txt; : > $O
for cfg in "FG_JIT_TASKS=0" "FG_JIT_TASKS=1" "FG_JIT_TASKS=1 FG_JIT_TASK_INLINE=0" "FG_JIT_TASKS=1 FG_JIT_TASK_INLINE=64"; do
  echo "---- $cfg" >> $O
  env $cfg timeout -k 10 500 python tools/bench_jit_all.py alldists logistic poisson_glm hier_logsigma hier_scale linreg mixture refmodel8 hier 2>&1 | grep -v amdgpu.ids | sed 's/  MH .*//' >> $O || exit 1
done
cat $O
