#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r4_mh_waves_8192.txt; : > $O
for w in 16 8 4; do
  echo "---- FG_HMC_WAVES=$w" >> $O
  FG_HMC_WAVES=$w timeout -k 10 200 python tools/bench_mh_phases.py 2>&1 | grep -v amdgpu.ids >> $O || exit 1
done
cat $O
