"""MH tiles of a CU started a part of a step apart (FG_MH_EXP bits 128: by tile number mod 4, 256: by quarter of the grid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fugue_amd import engine as E, workloads as W
data, _ = W.mixture_data(64)
for name, prog, C in (("refmodel20", W.reference_model(20), 65536), ("c5", W.mixture(data), 262144), ("normal32", W.normal_sites(32), 65536)):
    cp = E.compile_model(prog)
    for exp in (0, 128, 256):
        if exp: os.environ["FG_MH_EXP"] = str(exp)
        else: os.environ.pop("FG_MH_EXP", None)
        os.environ["FG_JIT"] = "0"
        eng = E.Engine(cp, C, seed=1)
        eng.mh_init(200); eng.mh_step(200); eng.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); eng.mh_step(400); eng.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"{name:12s} C={C:6d} exp={exp:3d} {eng.mh_last_kernel():22s} {C * 400 / np.median(ts):.3e} chain-steps/s", flush=True)
        eng.close()
