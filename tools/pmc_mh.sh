#!/bin/bash
# PMC passes over the MH kernel (counters in their own runs, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export FG_MH_ONLY=1
CMD="python3 $R/tools/bench_mh.py"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/mh_pmc1 -- $CMD > $R/gpurun_out/mh_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/mh_pmc2 -- $CMD > $R/gpurun_out/mh_pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM SQ_INST_CYCLES_SMEM --output-format csv -d $R/gpurun_out/mh_pmc3 -- $CMD > $R/gpurun_out/mh_pmc3.log 2>&1
cd $R && python3 tools/pmc_summarize.py gpurun_out/mh_pmc1 gpurun_out/mh_pmc2 gpurun_out/mh_pmc3
