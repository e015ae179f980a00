#!/bin/bash
# timing-only experiment builds of the gradient-stream HMC kernel (results are wrong by construction): which part of a
# transition costs what.  Every variant is built to its OWN library path (FG_LIB_PATH): the product library is never touched.
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/exp_libs
for defs in "" "FG_EXP_NOSTREAM" "FG_EXP_NOSTREAM,FG_EXP_NOSCORE" "FG_EXP_NOSTREAM,FG_EXP_NOSCORE,FG_EXP_NOMOM" "FG_EXP_NOSTREAM,FG_EXP_NOSCORE,FG_EXP_NOMOM,FG_EXP_NODA"; do
  tag=$(echo "x$defs" | tr ',' '_')
  export FG_LIB_PATH=$R/gpurun_out/exp_libs/libfugue_amd_$tag.so
  FG_EXTRA_DEFS=$defs python fugue_amd/build.py --force > /dev/null 2>&1
  FG_HMC_SEP=0 FG_HMC_WAVES=4 python bench.py --steps 100 --warmup 0 --launch 25 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('defs=[$defs] launch_ms=%.3f' % j['roofline']['avg_launch_ms'])
"
done
unset FG_LIB_PATH
