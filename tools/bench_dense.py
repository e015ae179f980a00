"""hmc_chain in the dense mode (grad_log_joint verbatim) on the headline model: k_hmc_sep_steps<MASS, 1>."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E, workloads as W
for name, prog in (("normal32", W.normal_sites(32)), ("refmodel32", W.reference_model(32))):
    cp = E.compile_model(prog)
    for C in (65536, 8192):
        eng = E.Engine(cp, C, seed=1)
        eng.hmc_init(E.hmc_config(grad_mode=E.GRAD_FD_DENSE), 0)
        eng.hmc_step(10); eng.synchronize()
        n = 40
        t0 = time.perf_counter(); eng.hmc_step(n); eng.synchronize(); dt = time.perf_counter() - t0
        print(f"{name:12s} C={C:6d} {eng.hmc_last_kernel():34s} {C * n * 16 / dt:.3e} leapfrog-steps/s (dense)", flush=True)
        eng.close()
