"""C3 (BASELINE configs[2]): linear regression, 32 Normal coefficients, 1 024 synthetic observations, hmc_chain.
Times a few transitions of the general interpreter kernel; prints leapfrog-steps/s and the posterior-mean error of
a short run against the closed-form ridge posterior."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fugue_amd import engine as E, workloads as W

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=65536)
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--p", type=int, default=32)
ap.add_argument("--transitions", type=int, default=2)
ap.add_argument("--leapfrog", type=int, default=16)
ap.add_argument("--grad", default="fd_sparse")
ap.add_argument("--check", action="store_true")
a = ap.parse_args()
X, y, beta = W.ridge_data(a.n, a.p)
t0 = time.perf_counter(); cp = E.compile_model(W.ridge_regression(X, y)); t_c = time.perf_counter() - t0
eng = E.Engine(cp, a.chains, seed=3)
mode = E.GRAD_FD_SPARSE if a.grad == "fd_sparse" else E.GRAD_FD_DENSE
t0 = time.perf_counter(); eng.hmc_init(E.hmc_config(grad_mode=mode, n_leapfrog=a.leapfrog, init_step_size=0.004), 0); eng.synchronize(); t_i = time.perf_counter() - t0
eng.hmc_step(1); eng.synchronize()
t0 = time.perf_counter(); eng.hmc_step(a.transitions); eng.synchronize(); dt = time.perf_counter() - t0
print(f"C3 n={a.n} p={a.p} chains={a.chains} {a.grad}: compile {t_c:.2f}s ({cp.n_instructions} instr, {sum(cp.dep_counts)} sub-program instr), "
      f"init {t_i:.2f}s, {a.chains * a.transitions * a.leapfrog / dt:.3e} leapfrog-steps/s ({dt / a.transitions * 1e3:.1f} ms/transition), accept {eng.hmc_stats().accept_rate:.3f}", flush=True)
if a.check:
    nw, ns = 150, 100
    eng2 = E.Engine(cp, 4096, seed=5)
    d = eng2.device_alloc(ns * cp.d * 4096 * 8)
    eng2.hmc_run(E.hmc_config(grad_mode=mode, n_leapfrog=a.leapfrog), ns, nw, d)
    draws = eng2.download(d, (ns, cp.d, 4096))
    mu, Sig = W.ridge_truth(X, y)
    order = [cp.site_names.index(f"beta#{j}") for j in range(a.p)]
    print("posterior mean max abs err", float(np.abs(draws.mean(axis=(0, 2))[order] - mu).max()), "accept", eng2.hmc_stats().accept_rate)
