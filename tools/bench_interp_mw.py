"""hmc_chain throughput of interpreter-bound models (no record stream): the one-wave kernel (FG_HMC_INTERP_MW=0) against
k_hmc_interp_mw_steps at W waves per tile and 2 / 3 / 4 waves per SIMD, at 65 536 and 8 192 chains.
usage: python tools/bench_interp_mw.py [model ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E
from tests.models import ZOO
L = 16
for name in sys.argv[1:] or ["alldists", "poisson_glm", "hier_logsigma", "logistic100"]:
    from fugue_amd import workloads as W
    cp = E.compile_model(W.logistic_regression(*W.classification_data(100)[:2]) if name == "logistic100" else ZOO[name]())
    for C in (65536, 8192):
        for jit, mw, occ, W in [(0, 0, 4, 0), (0, 1, 4, 0), (1, 1, 4, 0), (1, 1, 4, 2), (1, 1, 4, 4), (1, 1, 4, 16)]:
            os.environ["FG_JIT"] = str(jit)                  # 1: the model compiled at run time (fg_jit.cpp)
            os.environ["FG_HMC_INTERP_MW"] = str(mw)
            os.environ["FG_HMC_INTERP_OCC"] = str(occ)
            if W: os.environ["FG_HMC_INTERP_WAVES"] = str(W)
            else: os.environ.pop("FG_HMC_INTERP_WAVES", None)
            eng = E.Engine(cp, C, seed=2)
            eng.hmc_init(E.hmc_config(grad_mode=E.GRAD_FD_SPARSE, n_leapfrog=L), 10)
            eng.hmc_step(10); eng.synchronize()
            n = 10
            t0 = time.perf_counter(); eng.hmc_step(n); eng.synchronize(); dt = time.perf_counter() - t0
            print(f"{name:14s} d={cp.d:3d} C={C:6d} {eng.hmc_last_kernel():38s} {C * n * L / dt:.3e} leapfrog-steps/s", flush=True)
            eng.close()
