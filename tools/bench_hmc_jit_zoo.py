"""hmc_chain on programs compiled at run time (k_hmc_jit_steps), 65 536 and 8 192 chains.  usage: python tools/bench_hmc_jit_zoo.py [model ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E, workloads as W
from tests.models import ZOO
L = 16
for name in sys.argv[1:] or ["alldists", "poisson_glm", "hier_logsigma", "logistic100", "hier_scale"]:
    cp = E.compile_model(W.logistic_regression(*W.classification_data(100)[:2]) if name == "logistic100" else ZOO[name]())
    for C in (65536, 8192):
        eng = E.Engine(cp, C, seed=2)
        eng.hmc_init(E.hmc_config(grad_mode=E.GRAD_FD_SPARSE, n_leapfrog=L), 10)
        eng.hmc_step(10); eng.synchronize()
        n = 20
        t0 = time.perf_counter(); eng.hmc_step(n); eng.synchronize(); dt = time.perf_counter() - t0
        print(f"{name:14s} d={cp.d:3d} C={C:6d} {eng.hmc_last_kernel():38s} {C * n * L / dt:.3e} leapfrog-steps/s", flush=True)
        eng.close()
