#!/bin/bash
# timing breakdown of k_mh_mw_steps by skipping phases (FG_MH_EXP bits: 1 terms, 2 finish, 4 propose, 8 random numbers)
cd ${GRAFT_REPO_ROOT:-.}
for m in ref c5; do for x in 0 1 2 4 8 15; do echo -n "exp=$x "; FG_MH_EXP=$x python tools/ab_mh.py $m 2>&1 | grep -v amdgpu.ids; done; done
