#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sep_kernel" > gpurun_out/r3_sephalf_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3_sephalf_tests.log
tail -4 gpurun_out/r3_sephalf_tests.log
grep -q "tests rc 0" gpurun_out/r3_sephalf_tests.log || exit 1
for ch in 4096 8192; do for cfg in 0:auto 1:4 1:8 1:16; do
  half=${cfg%:*}; w=${cfg#*:}
  if [ $w = auto ]; then unset FG_HMC_WAVES; else export FG_HMC_WAVES=$w; fi
  FG_HMC_SEP_HALF=$half timeout -k 10 300 python bench.py --chains $ch --no-extras --no-cpu-baseline --steps 200 --warmup 50 --repeats 3 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('half=$half W=$w chains=$ch', '%.4g' % j['value'], j['roofline']['kernel'], j['timed_regions']['value']['all'])" | tee -a gpurun_out/r3_sephalf.txt
done; done
