#!/bin/bash
# headline bench for 1, 2 and 4 waves per 64-chain tile of the multi-wave HMC kernel
for w in 1 2 4; do
  FG_HMC_WAVES=$w python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('waves/tile=$w  %.3e lf-steps/s  launch_ms=%.3f  mean_err=%.2e rhat=%.4f' % (j['value'], j['roofline']['avg_launch_ms'], j['check']['posterior_mean_max_abs_err'], j['check']['split_rhat_max']))
"
done
