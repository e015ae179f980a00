"""Generated phase B of the multi-wave MH kernel: sixteen statement segments dealt to the waves (FG_MH_NSEG=0) against one segment per
wave (FG_MH_NSEG=1), with the control wave's share in sixteenths (FG_MH_CTL16)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E, workloads as W
from tests.models import ZOO
cases = [("refmodel20", lambda: W.reference_model(20), (65536, 8192)), ("normal32", lambda: W.normal_sites(32), (65536, 8192)), ("refmodel50", lambda: W.reference_model(50), (65536,)),
         ("linreg", ZOO["linreg"], (65536,)), ("hier_scale", ZOO["hier_scale"], (65536,))]
for name, mk, sizes in cases:
    cp = E.compile_model(mk())
    for C in sizes:
        for nseg, ctl in (("0", "16"), ("1", "16"), ("1", "10"), ("1", "6"), ("1", "2")):
            os.environ["FG_MH_NSEG"] = nseg; os.environ["FG_MH_CTL16"] = ctl
            eng = E.Engine(cp, C, seed=1)
            r = {}
            for label, nw in (("adapting", 10 ** 6), ("sampling", 0)):
                eng.mh_init(nw); eng.mh_step(300); eng.synchronize()
                best = 0.0
                for _ in range(3):
                    t0 = time.perf_counter(); eng.mh_step(200); eng.synchronize(); best = max(best, C * 200 / (time.perf_counter() - t0))
                r[label] = best
            print(f"{name:11s} C={C:6d} nseg={nseg} ctl16={ctl:>2s} {eng.mh_last_kernel()[:22]:22s} adapting {r['adapting']:.3e} sampling {r['sampling']:.3e}", flush=True)
            eng.close()
