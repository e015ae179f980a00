#!/bin/bash
# timing-only experiment builds (results are wrong by construction): which part of a gradient pass costs what
for defs in "" "FG_EXP_NODRIFT" "FG_EXP_NOSTREAM" "FG_EXP_NODRIFT,FG_EXP_NOSTREAM"; do
  FG_EXTRA_DEFS=$defs python fugue_amd/build.py --force > /dev/null 2>&1
  for L in 16 64; do
    python bench.py --steps 50 --warmup 0 --launch 25 --leapfrog $L --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('defs=[$defs] L=$L launch_ms=%.3f' % j['roofline']['avg_launch_ms'])
"
  done
done
