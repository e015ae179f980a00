#!/bin/bash
# round 4, first GPU call: the bench as the driver runs it (new compact line) + MH phase cycles at three chain counts
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_a.json 2> gpurun_out/r4_bench_a.err || { echo bench failed; tail -20 gpurun_out/r4_bench_a.err; exit 1; }
wc -c gpurun_out/r4_bench_a.json
cp gpurun_out/bench_full.json gpurun_out/r4_bench_a_full.json
export FG_LIB_PATH=$PWD/fugue_amd/lib/libfugue_prof.so FG_EXTRA_DEFS=FG_MH_PROF,FG_HMC_PROF FG_JIT=0
for c in 65536 16384 8192; do python tools/prof_mh_phases.py ref $c 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r4_mh_phases_before.txt
python tools/prof_mh_phases.py c5 32768 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4_mh_phases_before.txt
python tools/prof_hmc_phases.py 8192 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_hmc_phases_before.txt
echo ok
