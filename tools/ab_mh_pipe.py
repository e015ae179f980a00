"""A/B of the two step loops of the multi-wave MH kernel (FG_MH_PIPE=0: one control wave; 1: decider + speculative proposer), adapting and
sampling phases apart, HIP-event timed.  usage: python tools/ab_mh_pipe.py [ref|c5|all] [chain counts ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fugue_amd import engine as E, workloads as W

which = sys.argv[1] if len(sys.argv) > 1 else "ref"
counts = [int(x) for x in sys.argv[2:]] or [65536, 8192]
models = []
if which in ("ref", "all"): models.append(("refmodel20", lambda: W.reference_model(20)))
if which in ("c5", "all"): models.append(("c5", lambda: W.mixture(W.mixture_data(64)[0])))
for name, mk in models:
    cp = E.compile_model(mk())
    for C in counts:
        res = {}
        for pipe in os.environ.get("AB_PIPES", "0,1").split(","):
            os.environ["FG_MH_PIPE"] = pipe
            eng = E.Engine(cp, C, seed=1)
            rates = {}
            for label, nw in (("adapting", 10 ** 6), ("sampling", 0)):
                eng.mh_init(nw); eng.mh_step(400); eng.synchronize()
                best = 0.0
                for _ in range(3):
                    t0 = time.perf_counter(); eng.mh_step(300); eng.synchronize(); dt = time.perf_counter() - t0
                    best = max(best, C * 300 / dt)
                rates[label] = best
            res[pipe] = (rates, eng.mh_last_kernel(), eng.mh_stats().accept_rate)
            eng.close()
        for pipe, (rates, k, acc) in res.items():
            print(f"{name:10s} C={C:6d} pipe={pipe} {k:24s} adapting {rates['adapting']:.3e} sampling {rates['sampling']:.3e} accept {acc:.4f}", flush=True)
