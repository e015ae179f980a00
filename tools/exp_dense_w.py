import os, sys, time
sys.path.insert(0, '/root/repo')
from fugue_amd import engine as E, workloads as W
cp = E.compile_model(W.normal_sites(32))
for Wv in (0, 4, 8, 16):
    if Wv: os.environ["FG_HMC_WAVES"] = str(Wv)
    else: os.environ.pop("FG_HMC_WAVES", None)
    for C in (65536, 32768, 16384):
        eng = E.Engine(cp, C, seed=1)
        eng.hmc_init(E.hmc_config(grad_mode=E.GRAD_FD_DENSE), 0)
        eng.hmc_step(10); eng.synchronize()
        n = 40
        t0 = time.perf_counter(); eng.hmc_step(n); eng.synchronize(); dt = time.perf_counter() - t0
        print(f"forced W={Wv:2d} C={C:6d} {eng.hmc_last_kernel():34s} {C * n * 16 / dt:.3e} leapfrog-steps/s (dense)", flush=True)
        eng.close()
