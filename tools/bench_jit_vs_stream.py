"""hmc_chain: the hand-written stream kernels against the same program compiled at run time (FG_JIT=2 forces the compiled form)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fugue_amd import engine as E
from tests.models import ZOO
L = 16
for name in sys.argv[1:] or ["hier_scale", "mixture", "linreg", "hier", "refmodel8", "ridge8"]:
    cp = E.compile_model(ZOO[name]())
    res = {}
    for C in (65536, 8192):
        for jit in (1, 2):
            os.environ["FG_JIT"] = str(jit)
            eng = E.Engine(cp, C, seed=2)
            eng.hmc_init(E.hmc_config(grad_mode=E.GRAD_FD_SPARSE, n_leapfrog=L), 10)
            eng.hmc_step(10); eng.synchronize()
            n = 10
            t0 = time.perf_counter(); eng.hmc_step(n); eng.synchronize(); dt = time.perf_counter() - t0
            res[jit] = (eng.get_values(), eng.hmc_step_sizes())
            print(f"{name:12s} d={cp.d:3d} C={C:6d} {eng.hmc_last_kernel():44s} {C * n * L / dt:.3e} leapfrog-steps/s", flush=True)
            eng.close()
        print("   bit-identical:", all(np.array_equal(a, b, equal_nan=True) for a, b in zip(res[1], res[2])))
