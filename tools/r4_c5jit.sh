#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r4_c5_jit_any.txt; : > $O
for cfg in "FG_MH_JIT_ANY=0" "FG_MH_JIT_ANY=1 FG_MH_BAKE=2" "FG_MH_JIT_ANY=1 FG_MH_BAKE=0"; do
  echo "---- $cfg" >> $O
  env $cfg FG_JIT_VERBOSE=1 timeout -k 10 300 python tools/bench_c5.py 2>&1 | grep -v amdgpu.ids >> $O || exit 1
done
cat $O
