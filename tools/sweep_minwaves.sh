#!/bin/bash
# register budget of the interpreter kernels (FG_MIN_WAVES = waves per SIMD the VGPR count must allow) x chains per GPU:
# side measurements (dense FD, MH, SMC) of bench.py
for mw in 2 4; do
  FG_MIN_WAVES=$mw python fugue_amd/build.py --force > /dev/null 2>&1
  for ch in 65536 262144; do
    python bench.py --chains $ch --steps 50 --warmup 25 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); x = j['extras']
        print('min_waves=$mw chains=$ch  sparse %.3e  dense %.3e  mh %.3e  smc1M %.4f s' % (j['value'], x['hmc_fd_dense_leapfrog_steps_per_sec'], x['mh_chain_steps_per_sec'], x['smc_1m_particles_seconds']))
"
  done
done
python fugue_amd/build.py --force > /dev/null 2>&1
