"""Does a second resident tile per CU help the mixture (C5) MH step?  mixture(n) for n around the LDS boundary of two tiles per CU
(rows = (4 + n + 1) + (4 + 2 n) + 17 <= 159), 262 144 chains."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E, workloads as W
for n in (40, 44, 46, 48, 64):
    data, _ = W.mixture_data(n)
    cp = E.compile_model(W.mixture(data))
    for Wv in (0, 8, 16):
        if Wv: os.environ["FG_HMC_WAVES"] = str(Wv)
        else: os.environ.pop("FG_HMC_WAVES", None)
        eng = E.Engine(cp, 262144, seed=1)
        eng.mh_init(200); eng.mh_step(200); eng.synchronize()
        t0 = time.perf_counter(); eng.mh_step(200); eng.synchronize(); dt = time.perf_counter() - t0
        rows = (cp.S + 1) + (cp.S + cp.O) + 17
        print(f"mixture({n}) rows {rows} ({rows * 512 / 1024:.0f} KB) forced W={Wv:2d} {eng.mh_last_kernel()[:22]:22s} {262144 * 200 / dt:.3e} chain-steps/s  x statements = {262144 * 200 / dt * (cp.S + cp.O):.3e}", flush=True)
        eng.close()
