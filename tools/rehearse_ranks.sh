#!/bin/bash
# Rehearsal of the N > 1 bench path on a ONE-GPU box (not a measurement): two ranks drive cuda:0, collectives over gloo.
#   pass 1: every leg (HMC, MH, SMC, extras, validity) -- catches rank-asymmetric collectives (a rank-0-only leg that enters an
#           all-gather deadlocks here exactly as it would on 8 GPUs);
#   pass 2: the library's RCCL path forced on -- RCCL refuses two ranks on one device, so this exercises the failure path: every
#           rank must fall back to torch.distributed together.
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out
export FG_BENCH_ONE_DEVICE=1
timeout -k 10 240 python bench.py --gpus 2 --chains 16384 --steps 100 --warmup 50 > gpurun_out/rehearse_all.out 2> gpurun_out/rehearse_all.err || { echo "pass 1 FAILED"; tail -5 gpurun_out/rehearse_all.err; exit 1; }
grep "bench rank" gpurun_out/rehearse_all.err
FG_BENCH_FORCE_NATIVE_RCCL=1 FG_BENCH_RCCL_TIMEOUT=25 timeout -k 10 240 python bench.py --gpus 2 --chains 16384 --steps 100 --warmup 50 --no-extras > gpurun_out/rehearse_rccl.out 2> gpurun_out/rehearse_rccl.err || { echo "pass 2 FAILED"; tail -5 gpurun_out/rehearse_rccl.err; exit 1; }
python - <<'PY'
import json
for f in ("gpurun_out/rehearse_all.out", "gpurun_out/rehearse_rccl.out"):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    assert j["n_gpus"] == 2 and j["check"]["chains_in_rhat"] == 2 * 16384, j
    print(f, "ok:", j["check"]["diagnostics_path"])
PY
