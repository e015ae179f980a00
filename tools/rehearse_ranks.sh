#!/bin/bash
# Rehearsal of the N > 1 bench path on a ONE-GPU box (not a measurement): two ranks drive cuda:0, collectives over gloo.
#   pass 1: every leg (HMC, MH, SMC, extras, validity) -- catches rank-asymmetric collectives (a rank-0-only leg that enters an
#           all-gather deadlocks here exactly as it would on 8 GPUs);
#   pass 2: the library's RCCL path forced on -- RCCL refuses two ranks on one device and RETURNS an error: the ranks agree on it
#           and take the same O(d) exchange through torch.distributed; complete line, all chains in R-hat, exit code 0;
#   pass 3: a communicator that never forms (FG_BENCH_FAKE_RCCL_HANG): a bounded wait, then every rank finishes from its own
#           chains WITHOUT another collective, rank 0 still prints its line (marked failed) and the run exits non-zero.
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out
export FG_BENCH_ONE_DEVICE=1
timeout -k 10 240 python bench.py --gpus 2 --chains 16384 --steps 100 --warmup 50 > gpurun_out/rehearse_all.out 2> gpurun_out/rehearse_all.err || { echo "pass 1 FAILED"; tail -5 gpurun_out/rehearse_all.err; exit 1; }
grep "bench rank" gpurun_out/rehearse_all.err
FG_BENCH_FORCE_NATIVE_RCCL=1 FG_BENCH_RCCL_TIMEOUT=25 timeout -k 10 240 python bench.py --gpus 2 --chains 16384 --steps 100 --warmup 50 --no-extras > gpurun_out/rehearse_rccl.out 2> gpurun_out/rehearse_rccl.err
RC2=$?
echo "pass 2 exit code $RC2 (an exchange that returned an error falls back to torch.distributed: 0)"
FG_BENCH_FORCE_NATIVE_RCCL=1 FG_BENCH_FAKE_RCCL_HANG=1 FG_BENCH_RCCL_TIMEOUT=25 timeout -k 10 240 python bench.py --gpus 2 --chains 16384 --steps 100 --warmup 50 --no-extras > gpurun_out/rehearse_hang.out 2> gpurun_out/rehearse_hang.err
RC3=$?
echo "pass 3 exit code $RC3 (an exchange that did not return must exit non-zero)"
[ $RC3 -ne 0 ] || { echo "pass 3 FAILED: a hung RCCL path exited 0"; exit 1; }
export RC2
python - <<'PY'
import json
j = json.loads(open("gpurun_out/rehearse_all.out").read().strip().splitlines()[-1])
assert j["n_gpus"] == 2 and j["check"]["chains_in_rhat"] == 2 * 16384, j["check"]
assert 0 < j["check"]["diagnostics_exchange_bytes_per_rank"] < 1 << 20, j["check"]          # O(d) doubles, not O(chains)
print("pass 1 ok:", j["check"]["diagnostics_path"], "| exchanged", j["check"]["diagnostics_exchange_bytes_per_rank"], "B per rank")
assert "validity" in j
for leg in ("mh", "smc", "c3_65536", "c3_8192", "c5", "hmc_fd_dense"):
    assert leg in j["legs"] and j["legs"][leg]["value"] > 0, leg
assert len(open("gpurun_out/rehearse_all.out").read().strip().splitlines()[-1]) <= 6144, "the line must fit the driver's tail"
import os
j = json.loads(open("gpurun_out/rehearse_rccl.out").read().strip().splitlines()[-1])
p = j["check"]["diagnostics_path"]
if os.environ["RC2"] == "0":
    assert "returned an error" in p and j["check"]["chains_in_rhat"] == 2 * 16384, j["check"]
else:                                                  # this RCCL build hung instead of refusing: the watchdog path, as in pass 3
    assert p.startswith("failed") and j["check"]["chains_in_rhat"] == 16384, j["check"]
print("pass 2 ok:", p[:140])
j = json.loads(open("gpurun_out/rehearse_hang.out").read().strip().splitlines()[-1])
assert j["check"]["diagnostics_path"].startswith("failed") and j["check"]["chains_in_rhat"] == 16384 and "legs" not in j, j["check"]
print("pass 3 ok:", j["check"]["diagnostics_path"][:90])
PY
# pass 4: strong scaling -- the job is 65 536 chains IN TOTAL (BASELINE configs 3 and 5 sharded): every leg's whole-job rate, two ranks
timeout -k 10 300 python bench.py --gpus 2 --scaling strong --steps 100 --warmup 50 --no-cpu-baseline > gpurun_out/rehearse_strong.out 2> gpurun_out/rehearse_strong.err || { echo "pass 4 FAILED"; tail -5 gpurun_out/rehearse_strong.err; exit 1; }
python - <<'PY'
import json
j = json.loads(open("gpurun_out/rehearse_strong.out").read().strip().splitlines()[-1])
assert j["scaling"] == "strong" and j["n_gpus"] == 2 and j["config"]["chains_per_gpu"] == 32768 and j["config"]["chains_total"] == 65536, j["config"]
print("pass 4 ok (strong, 2 ranks on ONE device: rates are not measurements): headline %.3g" % j["value"],
      " ".join("%s %.3g" % (k, v["value"]) for k, v in j["legs"].items()))
PY
