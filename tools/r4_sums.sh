#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_mh.py tests/test_gpu_jit.py -x -q 2>&1 | tail -3 || exit 1
O=gpurun_out/r4_mh_sums_form.txt; : > $O
for cfg in "FG_MH_SUMS_FORM=0 FG_MH_BAKE=0" "FG_MH_BAKE=0" "FG_MH_SUMS_FORM=0" "" "FG_MH_SUMS_FORM=6"; do
  echo "---- $cfg" >> $O
  env $cfg timeout -k 10 200 python tools/bench_mh_phases.py 2>&1 | grep -v amdgpu.ids >> $O || exit 1
done
cat $O
