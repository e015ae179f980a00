"""adaptive_mcmc_chain on the reference's bench model (benches/f_perf.rs:78-109), 65 536 chains: chain-steps/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E, workloads as W
C = int(os.environ.get("FG_CHAINS", 65536))
eng = E.Engine(E.compile_model(W.reference_model(20)), C, seed=1)
eng.mh_init(100)
eng.mh_step(100); eng.synchronize()
t0 = time.perf_counter(); eng.mh_step(400); eng.synchronize(); dt = time.perf_counter() - t0
print(f"mh {C} chains: {C * 400 / dt:.3e} chain-steps/s, accept {eng.mh_stats().accept_rate:.3f}", flush=True)
import numpy as np
if os.environ.get("FG_MH_ONLY"): sys.exit(0)
data, _ = W.mixture_data(32)
C5 = int(os.environ.get("FG_CHAINS_C5", 262144))
eng = E.Engine(E.compile_model(W.mixture(data)), C5, seed=1)
eng.mh_init(200)
eng.mh_step(200); eng.synchronize()
t0 = time.perf_counter(); eng.mh_step(200); eng.synchronize(); dt = time.perf_counter() - t0
print(f"mh C5 mixture(32 obs, K=4) {C5} chains: {C5 * 200 / dt:.3e} chain-steps/s, accept {eng.mh_stats().accept_rate:.3f}", flush=True)
