#!/bin/bash
# what the terms phase of k_mh_mw_steps waits for: libraries whose fg_mh.hip was built with the record fetch / the operand
# reads short-circuited (-DFG_EXP_MH_NOFETCH / -DFG_EXP_MH_NOLDS; timing only, results are wrong), prepared under
# fugue_amd/lib/exp/ by swapping that one object
R=${GRAFT_REPO_ROOT:-.}
cd $R
for m in ref c5; do
  echo -n "product      "; python tools/ab_mh.py $m 2>&1 | grep -v amdgpu.ids
  for v in NOFETCH NOLDS BOTH; do echo -n "$v "; FG_LIB_PATH=$R/fugue_amd/lib/exp/libfugue_amd_$v.so python tools/ab_mh.py $m 2>&1 | grep -v amdgpu.ids; done
done
