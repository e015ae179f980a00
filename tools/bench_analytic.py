"""FG_GRAD_ANALYTIC (forward-mode derivative in the compiled unit) against FG_GRAD_FD_SPARSE on programs beyond Normal force terms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E
from tests.models import ZOO
for name in ("alldists", "logistic", "poisson_glm", "hier_logsigma", "hier_scale", "mixture"):
    cp = E.compile_model(ZOO[name]())
    for C in (65536,):
        row = f"{name:14s} C={C}"
        for mode, lab in ((E.GRAD_FD_SPARSE, "fd_sparse"), (E.GRAD_ANALYTIC, "analytic")):
            eng = E.Engine(cp, C, seed=1)
            eng.hmc_init(E.hmc_config(grad_mode=mode), 10); eng.hmc_step(10); eng.synchronize()
            t0 = time.perf_counter(); eng.hmc_step(20); eng.synchronize(); dt = time.perf_counter() - t0
            row += f"  {lab} {C * 20 * 16 / dt:.3e}"
            eng.close()
        print(row, flush=True)
