"""Late start of a CU's second tile (FG_HMC_STAGGER: 1 the second half of the grid, 2 the odd tiles): k_hmc_sep_steps at small chain counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fugue_amd import engine as E, workloads as W
cp = E.compile_model(W.normal_sites(32))
for half, C in ((2, 8192), (1, 16384), (0, 32768), (0, 65536)):
    os.environ["FG_HMC_SEP_HALF"] = str(half)
    for stg in (0, 1, 2):
        os.environ["FG_HMC_STAGGER"] = str(stg)
        eng = E.Engine(cp, C, seed=1)
        d = eng.device_alloc(25 * cp.d * C * 8)
        eng.hmc_init(E.hmc_config(), 25); eng.hmc_step(25); eng.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(8): eng.hmc_step(25, d)
            eng.synchronize(); ts.append((time.perf_counter() - t0) / 8)
        print(f"split={half} chains={C:6d} stagger={stg} {eng.hmc_last_kernel():38s} {np.median(ts) * 1e3:.3f} ms per launch  {C * 25 * 16 / np.median(ts):.3e}", flush=True)
        eng.close()
