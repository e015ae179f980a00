"""adaptive_mcmc_chain on programs without a score stream: the statement-segment kernel compiled at run time (k_mh_jit_steps,
FG_MH_NOSTREAM_MW=0) against the pipelined multi-wave kernel around the same generated statements (k_mh_mw_jit_steps, default)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fugue_amd import engine as E, workloads as W
from tests.models import ZOO
progs = [("alldists", ZOO["alldists"]()), ("poisson_glm", ZOO["poisson_glm"]()), ("hier_logsigma", ZOO["hier_logsigma"]()),
         ("logistic100", W.logistic_regression(*W.classification_data(100)[:2]))]
for name, prog in progs:
    cp = E.compile_model(prog)
    for C in (65536, 8192):
        res = {}
        for mw in (0, 1):
            os.environ["FG_MH_NOSTREAM_MW"] = str(mw)
            eng = E.Engine(cp, C, seed=2)
            eng.mh_init(100)
            eng.mh_step(100); eng.synchronize()
            t0 = time.perf_counter(); eng.mh_step(200); eng.synchronize(); dt = time.perf_counter() - t0
            res[mw] = (eng.get_values(), eng.mh_scales(), eng.mh_log_weight())
            print(f"{name:14s} C={C:6d} {eng.mh_last_kernel()[:40]:40s} {C * 200 / dt:.3e} chain-steps/s", flush=True)
            eng.close()
        print("   bit-identical:", all(np.array_equal(a, b, equal_nan=True) for a, b in zip(res[0], res[1])))
