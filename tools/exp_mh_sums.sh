#!/bin/bash
# control-wave in-order sums: rows per chunk of the pipelined two-chain form (experiment builds) + the product build
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_mh.py tests/test_gpu_parity.py -x -q -k "mh" > gpurun_out/r3_mh_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3_mh_tests.log
tail -3 gpurun_out/r3_mh_tests.log
grep -q "tests rc 0" gpurun_out/r3_mh_tests.log || exit 1
for ch in 2 4 8; do
  L=$PWD/gpurun_out/exp_lib_mhsum_$ch.so
  FG_LIB_PATH=$L FG_EXTRA_DEFS=FG_MH_SUM_CH=$ch python -c "from fugue_amd import build; build.build()" || exit 1
  for m in ref c5; do echo -n "CH=$ch "; FG_LIB_PATH=$L python tools/ab_mh.py $m 2>&1 | grep -v amdgpu.ids; done
done 2>&1 | tee gpurun_out/exp_mh_sums.txt
for sp in 0 1; do for m in ref c5; do echo -n "product SPLIT=$sp "; FG_MH_SPLIT=$sp python tools/ab_mh.py $m 2>&1 | grep -v amdgpu.ids; done; done 2>&1 | tee -a gpurun_out/exp_mh_sums.txt
