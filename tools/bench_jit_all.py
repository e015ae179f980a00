"""Round 4: every program class that runs through code generated at run time, HMC (leapfrog-steps/s) and MH (chain-steps/s, sampling),
65 536 and 8 192 chains -- to be compared with DESIGN 3.11's round-3 table (the generated functions read their tile through
generic pointers then: FLAT accesses)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E, workloads as W
from tests.models import ZOO
which = sys.argv[1:] or ["alldists", "logistic", "poisson_glm", "hier_logsigma", "hier_scale", "linreg", "mixture", "refmodel8"]
for name in which:
    cp = E.compile_model(ZOO[name]() if name in ZOO else W.reference_model(int(name[8:]) if name.startswith("refmodel") else 20))
    for C in (65536, 8192):
        eng = E.Engine(cp, C, seed=1)
        out = f"{name:14s} C={C:6d}"
        if cp.d > 0:
            eng.hmc_init(E.hmc_config(), 10); eng.hmc_step(10); eng.synchronize()
            t0 = time.perf_counter(); eng.hmc_step(20); eng.synchronize(); dt = time.perf_counter() - t0
            out += f"  HMC {C * 20 * 16 / dt:.3e} [{eng.hmc_last_kernel()[:28]}]"
        eng.mh_init(200); eng.mh_step(300); eng.synchronize()
        t0 = time.perf_counter(); eng.mh_step(400); eng.synchronize(); dt = time.perf_counter() - t0
        out += f"  MH {C * 400 / dt:.3e} [{eng.mh_last_kernel()[:30]}]"
        print(out, flush=True)
        eng.close()
