"""hmc_chain on the north-star model with the reference-verbatim dense finite difference (FG_GRAD_FD_DENSE), 65 536 chains."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E, workloads as W
cp = E.compile_model(W.normal_sites(32))
eng = E.Engine(cp, 65536, seed=1)
eng.hmc_init(E.hmc_config(grad_mode=E.GRAD_FD_DENSE), 0)
eng.hmc_step(10); eng.synchronize()
t0 = time.perf_counter(); eng.hmc_step(50); eng.synchronize(); dt = time.perf_counter() - t0
print("dense %.3e leapfrog-steps/s" % (65536 * 50 * 16 / dt))
