#!/bin/bash
# A/B of the HMC kernels on the headline workload (one process per variant; same box, back to back)
R=${GRAFT_REPO_ROOT:-.}
cd $R
VARIANTS=${VARIANTS:-"FG_HMC_SEP=0;FG_HMC_SEP=1;FG_HMC_SEP=1 FG_HMC_WAVES=8;FG_HMC_SEP=1 FG_HMC_WAVES=4"}
CHAINS=${CHAINS:-"65536 16384"}
IFS=';' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do
  for ch in $CHAINS; do
    echo "== $v chains=$ch"
    env $v python3 bench.py --steps 200 --warmup 50 --chains $ch --no-extras --no-cpu-baseline 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        j = json.loads(l); print('value %.3e  launch_ms %.3f  mean_err %.2e  rhat %.4f  accept %.3f' % (j['value'], j['roofline']['avg_launch_ms'], j['check']['posterior_mean_max_abs_err'], j['check']['split_rhat_max'], j['check']['accept_rate']))
    elif l and 'amdgpu.ids' not in l: print(l[:200])
"
  done
done
