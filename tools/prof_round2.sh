#!/bin/bash
# Every profile committed under profiles/round2_*: kernel stats of the default bench command, HBM traffic of the three hot kernels
# (FETCH_SIZE / WRITE_SIZE in separate --pmc passes), instruction mixes of the HMC and MH kernels, SMC kernel stats.
# Counters always in their own rocprofv3 runs with --kernel-trace only; the program sits directly after `--`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2prof
rm -rf $O && mkdir -p $O
set -e
# 1. the default bench command, as the driver runs it
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 $R/bench.py > $O/bench.json 2> $O/bench.log
echo "[prof] bench stats done"
# 2. HBM traffic: HMC headline kernel
HMC="python3 $R/bench.py --steps 100 --warmup 50 --launch 25 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/hmc_fetch -- $HMC > $O/hmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/hmc_write -- $HMC > $O/hmc_write.log 2>&1
echo "[prof] hmc traffic done"
# 3. HBM traffic: MH (reference_model(20) + C5) and SMC
export FG_CHAINS_C5=262144
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/mh_fetch -- python3 $R/tools/bench_mh.py > $O/mh_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/mh_write -- python3 $R/tools/bench_mh.py > $O/mh_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/smc_fetch -- python3 $R/tools/bench_smc.py > $O/smc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/smc_write -- python3 $R/tools/bench_smc.py > $O/smc_write.log 2>&1
echo "[prof] mh/smc traffic done"
# 4. instruction mix: HMC and MH kernels
for leg in hmc mh; do
  if [ $leg = hmc ]; then CMD="python3 $R/bench.py --steps 50 --warmup 25 --launch 25 --no-cpu-baseline --no-extras"; else CMD="python3 $R/tools/bench_mh.py"; fi
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $O/${leg}_pmc1 -- $CMD > $O/${leg}_pmc1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $O/${leg}_pmc2 -- $CMD > $O/${leg}_pmc2.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM SQ_INST_CYCLES_SMEM --output-format csv -d $O/${leg}_pmc3 -- $CMD > $O/${leg}_pmc3.log 2>&1
  (cd $R && python3 tools/pmc_summarize.py $O/${leg}_pmc1 $O/${leg}_pmc2 $O/${leg}_pmc3 > $O/${leg}_instruction_mix.txt)
  echo "[prof] $leg mix done"
done
# 5. SMC kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/smc -- python3 $R/tools/bench_smc.py > $O/smc.log 2>&1
cd $R && python3 tools/prof_round2_collect.py $O
