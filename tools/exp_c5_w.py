import os, sys, time
sys.path.insert(0, '/root/repo')
from fugue_amd import engine as E, workloads as W
for n in (44, 42):
    data, _ = W.mixture_data(n)
    cp = E.compile_model(W.mixture(data))
    for rep in range(2):
        for Wv in (0, 16, 8):
            if Wv: os.environ["FG_HMC_WAVES"] = str(Wv)
            else: os.environ.pop("FG_HMC_WAVES", None)
            eng = E.Engine(cp, 262144, seed=1)
            eng.mh_init(200); eng.mh_step(200); eng.synchronize()
            t0 = time.perf_counter(); eng.mh_step(200); eng.synchronize(); dt = time.perf_counter() - t0
            print(f"mixture({n}) rep {rep} forced W={Wv:2d} {eng.mh_last_kernel()[:22]:22s} {262144 * 200 / dt:.3e} chain-steps/s", flush=True)
            eng.close()
