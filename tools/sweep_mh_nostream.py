"""W sweep of the pipelined MH kernel on programs without a score stream (FG_HMC_WAVES forces W)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E, workloads as W
from tests.models import ZOO
progs = [("alldists", ZOO["alldists"]()), ("logistic100", W.logistic_regression(*W.classification_data(100)[:2])), ("poisson_glm", ZOO["poisson_glm"]())]
for name, prog in progs:
    cp = E.compile_model(prog)
    for C in (65536, 8192):
        for Wv in (0, 2, 4, 8, 16):
            if Wv: os.environ["FG_HMC_WAVES"] = str(Wv)
            else: os.environ.pop("FG_HMC_WAVES", None)
            eng = E.Engine(cp, C, seed=2)
            eng.mh_init(100)
            eng.mh_step(100); eng.synchronize()
            t0 = time.perf_counter(); eng.mh_step(200); eng.synchronize(); dt = time.perf_counter() - t0
            print(f"{name:14s} C={C:6d} forced W={Wv:2d} {eng.mh_last_kernel()[:24]:24s} {C * 200 / dt:.3e} chain-steps/s", flush=True)
            eng.close()
