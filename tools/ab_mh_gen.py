"""Phase B of the multi-wave MH kernel: hand-written record runs (FG_MH_GEN_ALL=0, the rule of round 3) against every statement generated
(FG_MH_GEN_ALL=1) -- adapting / sampling chain-steps/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E, workloads as W
from tests.models import ZOO
cases = [("refmodel20", lambda: W.reference_model(20), (65536, 16384, 8192)), ("c5", lambda: W.mixture(W.mixture_data(64)[0]), (262144, 32768)),
         ("normal32", lambda: W.normal_sites(32), (65536, 8192)), ("refmodel8", lambda: W.reference_model(8), (65536, 8192)),
         ("refmodel50", lambda: W.reference_model(50), (65536, 8192)), ("hier_scale", ZOO["hier_scale"], (65536, 8192)), ("linreg", ZOO["linreg"], (65536, 8192))]
for name, mk, sizes in cases:
    cp = E.compile_model(mk())
    for C in sizes:
        for g in ("0", "1"):
            os.environ["FG_MH_GEN_ALL"] = g
            eng = E.Engine(cp, C, seed=1)
            r = {}
            for label, nw in (("adapting", 10 ** 6), ("sampling", 0)):
                eng.mh_init(nw); eng.mh_step(300); eng.synchronize()
                best = 0.0
                for _ in range(3):
                    t0 = time.perf_counter(); eng.mh_step(200); eng.synchronize(); best = max(best, C * 200 / (time.perf_counter() - t0))
                r[label] = best
            print(f"{name:11s} C={C:6d} gen_all={g} {eng.mh_last_kernel()[:26]:26s} adapting {r['adapting']:.3e} sampling {r['sampling']:.3e}", flush=True)
            eng.close()
