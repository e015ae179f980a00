"""MH on reference_model(20) at small chain counts: the two in-order sums on one / two waves, waves per tile."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fugue_amd import engine as E, workloads as W
cp = E.compile_model(W.reference_model(20))
for C in (8192, 16384):
    for split in ("auto", 0, 1):
        for Wv in (0, 8):
            if split == "auto": os.environ.pop("FG_MH_SPLIT", None)
            else: os.environ["FG_MH_SPLIT"] = str(split)
            if Wv: os.environ["FG_HMC_WAVES"] = str(Wv)
            else: os.environ.pop("FG_HMC_WAVES", None)
            eng = E.Engine(cp, C, seed=1)
            eng.mh_init(200); eng.mh_step(200); eng.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); eng.mh_step(400); eng.synchronize(); ts.append(time.perf_counter() - t0)
            print(f"C={C:6d} split={split!s:4s} forced W={Wv:2d} {eng.mh_last_kernel():20s} {C * 400 / np.median(ts):.3e} chain-steps/s  ({np.median(ts) / 400 * 1e6:.2f} us per step)", flush=True)
            eng.close()
