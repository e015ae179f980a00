"""adaptive_mcmc_chain: the hand-written multi-wave stream kernel (FG_JIT=0) against the same kernel with its statements compiled
at run time (default) and the statement-segment kernel compiled at run time (FG_JIT=2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fugue_amd import engine as E, workloads as W
from tests.models import ZOO
data, _ = W.mixture_data(64)
for name, prog, sizes in [("refmodel20", W.reference_model(20), (65536, 8192)), ("c5 mixture", W.mixture(data), (262144, 16384)), ("hier_scale", ZOO["hier_scale"](), (65536, 8192)), ("linreg", ZOO["linreg"](), (65536, 8192)), ("normal32", W.normal_sites(32), (65536, 8192))]:
    cp = E.compile_model(prog)
    for C in sizes:
        res = {}
        for jit in (0, 1, 2):
            os.environ["FG_JIT"] = str(jit)
            eng = E.Engine(cp, C, seed=1)
            eng.mh_init(200)
            eng.mh_step(200); eng.synchronize()
            t0 = time.perf_counter(); eng.mh_step(400); eng.synchronize(); dt = time.perf_counter() - t0
            res[jit] = (eng.get_values(), eng.mh_scales())
            print(f"{name:12s} C={C:6d} {eng.mh_last_kernel()[:44]:44s} {C * 400 / dt:.3e} chain-steps/s (sampling)", flush=True)
            eng.close()
        print("   bit-identical:", all(np.array_equal(a, b, equal_nan=True) for a, b in zip(res[0], res[1])) and all(np.array_equal(a, b, equal_nan=True) for a, b in zip(res[0], res[2])))
