#!/bin/bash
# profiles/round2_other_workloads.txt: the parity-test models, C3 (65 536 chains and the 8 192 chains per GPU of its 8-GPU sharding), MH, SMC and a
# chain-count sweep of the headline
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/other_workloads.txt
: > $O
timeout -k 10 300 python tools/bench_models.py 2>&1 | grep "kernel)" >> $O
timeout -k 10 400 python tools/bench_c3.py --check 2>&1 | grep -v amdgpu.ids >> $O
timeout -k 10 300 python tools/bench_c3.py --chains 8192 2>&1 | grep -v amdgpu.ids >> $O
timeout -k 10 300 python tools/bench_mh.py 2>&1 | grep -v amdgpu.ids >> $O
timeout -k 10 300 python tools/ab_mh.py c5 2>&1 | grep -v amdgpu.ids >> $O
timeout -k 10 300 python tools/bench_smc.py 2>&1 | grep -v amdgpu.ids >> $O
timeout -k 10 300 python tools/bench_dense.py 2>&1 | grep -v amdgpu.ids >> $O
timeout -k 10 300 python tools/bench_readme.py 2>&1 | grep -v amdgpu.ids >> $O
for c in 4096 8192 16384 32768 65536 131072 262144; do
  timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --chains $c --steps 400 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hmc headline chains $c %.4g leapfrog-steps/s frac %.3f' % (d['value'], d['roofline']['frac']))" >> $O
done
cat $O
