#!/bin/bash
# timing-only experiment builds (results are wrong by construction): which part of a transition costs what
for defs in "" "FG_EXP_NOSTREAM" "FG_EXP_NOSTREAM,FG_EXP_NOSCORE" "FG_EXP_NOSTREAM,FG_EXP_NOSCORE,FG_EXP_NOMOM" "FG_EXP_NOSTREAM,FG_EXP_NOSCORE,FG_EXP_NOMOM,FG_EXP_NODA"; do
  FG_EXTRA_DEFS=$defs python fugue_amd/build.py --force > /dev/null 2>&1
  for w in 4; do
    FG_HMC_WAVES=$w python bench.py --steps 100 --warmup 0 --launch 25 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('defs=[$defs] waves=$w launch_ms=%.3f' % j['roofline']['avg_launch_ms'])
"
  done
done
python fugue_amd/build.py --force > /dev/null 2>&1
