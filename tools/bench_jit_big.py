"""Logistic regression (examples/classification.rs's model) with n observations: the compiled form (plates rolled into loops over a
constant table) against the multi-wave interpreter kernels.  usage: python tools/bench_jit_big.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fugue_amd import engine as E, workloads as W
C = 65536
JIT_ONLY = "--jit-only" in sys.argv                      # skip the interpreter kernels (minutes per run at 100 000 observations)
for n in [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [100, 1000, 10000]:
    X, y, _ = W.classification_data(n)
    t0 = time.perf_counter(); cp = E.compile_model(W.logistic_regression(X, y)); t_build = time.perf_counter() - t0
    res = {}
    for jit in ((1,) if JIT_ONLY else (0, 1)):
        os.environ["FG_JIT"] = str(jit)
        eng = E.Engine(cp, C, seed=1)
        t0 = time.perf_counter()
        eng.hmc_init(E.hmc_config(init_step_size=0.02), 2); eng.hmc_step(2); eng.synchronize()
        t_first = time.perf_counter() - t0
        k = 3 if n >= 10000 else 10
        t0 = time.perf_counter(); eng.hmc_step(k); eng.synchronize(); dt = time.perf_counter() - t0
        res[jit] = (eng.get_values(), eng.hmc_log_joint())
        print(f"n={n:6d} program built in {t_build:.1f} s  {eng.hmc_last_kernel():44s} first steps {t_first:5.2f} s   {C * k * 16 / dt:.3e} leapfrog-steps/s", flush=True)
        eng.mh_init(20); eng.mh_step(20); eng.synchronize()
        t0 = time.perf_counter(); eng.mh_step(40); eng.synchronize(); dt = time.perf_counter() - t0
        print(f"          {eng.mh_last_kernel():44s} {C * 40 / dt:.3e} MH chain-steps/s", flush=True)
        eng.close()
    if not JIT_ONLY: print("   HMC bit-identical:", all(np.array_equal(a, b, equal_nan=True) for a, b in zip(res[0], res[1])))
