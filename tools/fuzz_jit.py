"""Randomised bit-identity cross-check of the run-time compiled kernels against the interpreter kernels (HMC and MH) on
tests/random_models.py::random_expression_program(seed).  usage: python tools/fuzz_jit.py [first_seed] [count]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fugue_amd import engine as E
from tests.random_models import random_expression_program
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 100)
bad = 0
for seed in range(first, first + count):
    cp = E.compile_model(random_expression_program(seed))
    out = []
    for jit in (0, 1):
        os.environ["FG_JIT"] = str(jit)
        eng = E.Engine(cp, 96, seed=seed)
        eng.prior_init()
        eng.hmc_init(E.hmc_config(n_leapfrog=3, init_step_size=0.01), 4); eng.hmc_step(8)
        v = eng.get_values(); lj = eng.hmc_log_joint(); kh = eng.hmc_last_kernel()
        eng.mh_init(15); eng.mh_step(30)
        out.append((v, lj, eng.get_values(), eng.mh_scales(), eng.mh_log_weight(), kh, eng.mh_last_kernel()))
        eng.close()
    same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out[0][:5], out[1][:5]))
    compiled = out[1][5].startswith("k_hmc_jit") and out[1][6].startswith(("k_mh_jit", "k_mh_mw_jit"))
    if not same or not compiled:
        bad += 1
        print("seed", seed, "MISMATCH" if not same else "not compiled", out[1][5], out[1][6], flush=True)
print(f"{count} programs from seed {first}: {bad} problems", flush=True)
