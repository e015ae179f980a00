#!/bin/bash
# Profiles committed under profiles/<round>_* (FG_PROF_ROUND, default round4): per configuration, kernel-trace stats + five rocprofv3 --pmc passes (instruction
# counts, activity / wait cycles, f64 instruction mix, FETCH_SIZE, WRITE_SIZE -- counters always in their own runs with
# --kernel-trace only; the program sits directly after `--`).  usage: prof_round.sh <tag> <key> [<key> ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
export FG_PROF_ROUND=${FG_PROF_ROUND:-round4}
O=$R/gpurun_out/prof_${FG_PROF_ROUND}_$TAG
mkdir -p $O
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
P3="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM SQ_INST_CYCLES_SMEM"
for KEY in "$@"; do
  D=$O/$(echo "$KEY" | tr '|' '_')
  mkdir -p $D
  echo "$KEY" > $D/key.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $R/tools/prof_driver.py "$KEY" > $D/stats.log 2>&1 || { echo "[prof] $KEY stats FAILED"; tail -3 $D/stats.log; exit 1; }
  rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $D/pmc1 -- python3 $R/tools/prof_driver.py "$KEY" > $D/pmc1.log 2>&1 || { echo "[prof] $KEY pmc1 FAILED"; exit 1; }
  rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $D/pmc2 -- python3 $R/tools/prof_driver.py "$KEY" > $D/pmc2.log 2>&1 || { echo "[prof] $KEY pmc2 FAILED"; exit 1; }
  rocprofv3 --kernel-trace --pmc $P3 --output-format csv -d $D/pmc3 -- python3 $R/tools/prof_driver.py "$KEY" > $D/pmc3.log 2>&1 || { echo "[prof] $KEY pmc3 FAILED"; exit 1; }
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 $R/tools/prof_driver.py "$KEY" > $D/fetch.log 2>&1 || { echo "[prof] $KEY fetch FAILED"; exit 1; }
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 $R/tools/prof_driver.py "$KEY" > $D/write.log 2>&1 || { echo "[prof] $KEY write FAILED"; exit 1; }
  echo "[prof] $KEY done"
done
cd $R && python3 tools/prof_round_collect.py $O
