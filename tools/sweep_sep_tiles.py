"""Tile width of k_hmc_sep_steps on the headline model: full (64 chains), half (32), quarter (16) tiles at small chain counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fugue_amd import engine as E, workloads as W
cp = E.compile_model(W.normal_sites(32))
for C in (4096, 8192, 16384, 32768):
    for half in ("auto", 0, 1, 2):
        if half == "auto": os.environ.pop("FG_HMC_SEP_HALF", None)
        else: os.environ["FG_HMC_SEP_HALF"] = str(half)
        eng = E.Engine(cp, C, seed=1)
        d = eng.device_alloc(25 * cp.d * C * 8)
        eng.hmc_init(E.hmc_config(), 25); eng.hmc_step(25); eng.synchronize()
        rates = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(8): eng.hmc_step(25, d)
            eng.synchronize(); rates.append(C * 200 * 16 / (time.perf_counter() - t0))
        print(f"chains={C:6d} tiles={half!s:4s} {eng.hmc_last_kernel():40s} {np.median(rates):.3e} leapfrog-steps/s", flush=True)
        eng.close()
