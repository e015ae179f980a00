#!/bin/bash
# what the terms phase of k_mh_mw_steps waits for: fg_mh.hip rebuilt with the record fetch / the operand reads short-circuited
# (-DFG_EXP_MH_NOFETCH / -DFG_EXP_MH_NOLDS; timing only, results are wrong), linked with the product's other objects into
# libraries of their own under gpurun_out/exp_libs/ (FG_LIB_PATH selects them; the product library is never touched)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
ls fugue_amd/lib/obj/*.o > /dev/null 2>&1 || python fugue_amd/build.py --force > /dev/null    # the objects do not travel to the GPU box
O=gpurun_out/exp_libs; mkdir -p $O
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DFG_BUILD -Wno-unused-function -w"
OBJS=$(ls -t fugue_amd/lib/obj/*.o | awk -F/ '{split($NF,a,"."); if (!(a[1] in seen)) {seen[a[1]]=1; print}}' | grep -v "/fg_mh\.")
for v in NOFETCH NOLDS BOTH; do
  case $v in NOFETCH) D="-DFG_EXP_MH_NOFETCH";; NOLDS) D="-DFG_EXP_MH_NOLDS";; BOTH) D="-DFG_EXP_MH_NOFETCH -DFG_EXP_MH_NOLDS";; esac
  hipcc $FLAGS $D -c -x hip fugue_amd/csrc/fg_mh.hip -o $O/fg_mh_$v.o && hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $O/fg_mh_$v.o -o $O/libfugue_amd_$v.so || exit 1
done
for m in ref c5; do
  echo -n "product "; python tools/ab_mh.py $m 2>&1 | grep -v amdgpu.ids
  for v in NOFETCH NOLDS BOTH; do echo -n "$v "; FG_LIB_PATH=$R/$O/libfugue_amd_$v.so python tools/ab_mh.py $m 2>&1 | grep -v amdgpu.ids; done
done
