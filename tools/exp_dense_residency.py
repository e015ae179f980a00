"""Does a second resident tile per CU help the dense finite-difference mode (k_hmc_sep_steps<.., 1>)?  normal_sites(d) around the LDS
boundary of two tiles per CU (rows = 8 d + 3 <= 159: d <= 19), 65 536 chains; the work per leapfrog step grows as d^2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E, workloads as W
for d in (16, 18, 19, 20, 22, 32):
    cp = E.compile_model(W.normal_sites(d))
    eng = E.Engine(cp, 65536, seed=1)
    eng.hmc_init(E.hmc_config(grad_mode=E.GRAD_FD_DENSE), 0)
    eng.hmc_step(10); eng.synchronize()
    n = 40
    t0 = time.perf_counter(); eng.hmc_step(n); eng.synchronize(); dt = time.perf_counter() - t0
    r = 65536 * n * 16 / dt
    print(f"normal_sites({d}) rows {8 * d + 3} ({(8 * d + 3) * 512 / 1024:.0f} KB) {eng.hmc_last_kernel():30s} {r:.3e} leapfrog-steps/s  x d^2 = {r * d * d:.3e}", flush=True)
    eng.close()
