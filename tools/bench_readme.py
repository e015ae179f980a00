"""C2 as BASELINE.json words it: the README model (mu ~ N(0,1); y ~ N(mu, 0.5) = 1.2), hmc_chain, 65 536 chains x 1 000 steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fugue_amd import engine as E, workloads as W
C = int(os.environ.get("FG_CHAINS", 65536))
cp = E.compile_model(W.readme_normal())
eng = E.Engine(cp, C, seed=1)
d = eng.device_alloc(1000 * cp.d * C * 8)
eng.hmc_run(E.hmc_config(), 50, 50, d); eng.synchronize()
eng = E.Engine(cp, C, seed=1)
t0 = time.perf_counter(); st = eng.hmc_run(E.hmc_config(), 1000, 200, d); eng.synchronize(); dt = time.perf_counter() - t0
x = eng.download(d, (1000, cp.d, C))
print(f"README model, {C} chains, 200 warmup + 1000 sampling transitions (L = 16): {dt * 1e3:.1f} ms, {C * 1200 * 16 / dt:.3e} leapfrog-steps/s, "
      f"posterior mean {x.mean():.5f} (0.96), var {x.var():.5f} (0.2), accept {st.accept_rate:.3f}")
