#!/bin/bash
# the launch shape baked into the compiled MH units (FG_MH_BAKE) against kernel arguments, every compiled path
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r4_mh_bake.txt; : > $O
for cfg in "FG_MH_BAKE=0" "FG_MH_BAKE=1"; do
  echo "---- $cfg" >> $O
  env $cfg timeout -k 10 500 python tools/bench_jit_all.py 2>&1 | grep -v amdgpu.ids | sed 's/HMC [^ ]* \[[^]]*\]//' >> $O || exit 1
  env $cfg timeout -k 10 200 python tools/bench_mh_phases.py 2>&1 | grep -v amdgpu.ids >> $O || exit 1
done
cat $O
