#!/bin/bash
# builds the library with different register budgets and tile widths, benches each
set -e
for mw in 1 2 4; do
  FG_MIN_WAVES=$mw python fugue_amd/build.py --force > /dev/null 2>&1
  for tw in 64 32 16; do
    for g in fd_sparse fd_dense; do
      st=40; [ $g = fd_dense ] && st=4
      FG_TILE_WIDTH=$tw python bench.py --steps $st --warmup 10 --launch $st --grad $g --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('min_waves=$mw tile=$tw', j['config']['grad'], '%.3e lf/s' % j['value'], 'launch_ms=%.1f' % j['roofline']['avg_launch_ms'], 'mean_err=%.1e' % j['check']['posterior_mean_max_abs_err'])
"
    done
  done
done
