import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from fugue_amd import engine as E, workloads as W
cp = E.compile_model(W.reference_model(20))
C = 65536
for rep in range(3):
    for stg in (0, 1):
        os.environ["FG_MH_STAGGER"] = str(stg)
        eng = E.Engine(cp, C, seed=1)
        eng.mh_init(200); eng.mh_step(100); eng.synchronize()
        rates = []
        for _ in range(3):
            eng.mh_init(200); eng.synchronize()
            t0 = time.perf_counter()
            for _ in range(6): eng.mh_step(100)
            eng.synchronize(); rates.append(C * 600 / (time.perf_counter() - t0))
        print(f"stagger={stg} rep {rep}: 200 adapting + 400 sampling steps: {np.median(rates):.3e} chain-steps/s", flush=True)
        eng.close()
