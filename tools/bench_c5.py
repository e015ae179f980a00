"""MH on BASELINE's C5 (mixture: 4 components, 64 observations, 68 sites): chain-steps/s, adapting and sampling, 262 144 and 32 768 chains."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E, workloads as W
cp = E.compile_model(W.mixture(W.mixture_data(64)[0]))
for C in (262144, 32768):
    eng = E.Engine(cp, C, seed=1)
    out = []
    for label, nw in (("adapting", 1000000), ("sampling", 0)):
        eng.mh_init(nw); eng.mh_step(200); eng.synchronize()
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter(); eng.mh_step(200); eng.synchronize(); best = max(best, C * 200 / (time.perf_counter() - t0))
        out.append(f"{label} {best:.3e}")
    print(f"C5 C={C:6d} {eng.mh_last_kernel()[:24]:24s} " + "  ".join(out), flush=True)
    eng.close()
