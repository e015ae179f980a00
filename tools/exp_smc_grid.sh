#!/bin/bash
# ESS pass grid experiment: the same run with k_smc_ess2_pass launched as B blocks x T threads, U particles per trip (experiment builds, FG_LIB_PATH)
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
for cfg in 256x512x4 256x512x8 256x1024x4 128x1024x4 128x1024x8 512x256x8 512x512x4 64x1024x8; do
  B=${cfg%%x*}; r=${cfg#*x}; T=${r%x*}; U=${r#*x}
  L=$PWD/gpurun_out/exp_lib_ess2_$cfg.so
  FG_LIB_PATH=$L FG_EXTRA_DEFS=ESS2_BLOCKS=$B,ESS2_THREADS=$T,ESS2_UNROLL=$U python -c "from fugue_amd import build; build.build()" || exit 1
  echo -n "ESS2 $cfg: "; FG_LIB_PATH=$L python tools/bench_smc.py 2>&1 | grep "smc " | tail -2 | tr '\n' ' '; echo
done 2>&1 | tee gpurun_out/exp_smc_grid2.txt
