"""adaptive_mcmc_chain on reference_model(20): chain-steps/s of the adapting and of the sampling phase, 65 536 and 8 192 chains."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E, workloads as W
cp = E.compile_model(W.reference_model(20))
for C in (65536, 8192):
    eng = E.Engine(cp, C, seed=1)
    out = []
    for label, nw in (("adapting", 1000000), ("sampling", 0)):
        eng.mh_init(nw); eng.mh_step(200); eng.synchronize()
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter(); eng.mh_step(400); eng.synchronize(); dt = time.perf_counter() - t0
            best = max(best, C * 400 / dt)
        out.append(f"{label} {best:.3e}")
    print(f"refmodel20 C={C:6d} {eng.mh_last_kernel()[:28]:28s} " + "  ".join(out), flush=True)
    eng.close()
