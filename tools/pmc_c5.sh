#!/bin/bash
# PMC passes over k_mh_mw_steps<3> on C5 (tools/ab_mh.py c5): instruction counts, active / wait cycles, LDS
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c5pmc
rm -rf $O && mkdir -p $O
CMD="python3 $R/tools/ab_mh.py c5"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $O/p1 -- $CMD > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $O/p2 -- $CMD > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_INST_CYCLES_SMEM SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d $O/p3 -- $CMD > $O/p3.log 2>&1
cd $R && python3 tools/pmc_summarize.py $O/p1 $O/p2 $O/p3 > $O/c5_mix.txt; cat $O/c5_mix.txt
