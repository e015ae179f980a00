#!/bin/bash
# PMC passes over the general interpreter kernel on the C3 regression model
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python3 $R/tools/bench_c3.py --chains 65536 --transitions 1"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/c3_pmc1 -- $CMD > $R/gpurun_out/c3_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/c3_pmc2 -- $CMD > $R/gpurun_out/c3_pmc2.log 2>&1
cd $R && python3 tools/pmc_summarize.py gpurun_out/c3_pmc1 gpurun_out/c3_pmc2
