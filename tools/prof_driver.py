"""One profiled configuration per invocation (rocprofv3 sits in front: tools/prof_round.sh).  Every dispatch of the kernel under
study covers a known number of units (transitions / chain steps / runs); tools/prof_round_collect.py divides by them.
usage: python tools/prof_driver.py <key>      keys: see CONFIGS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fugue_amd import engine as E, workloads as W

# key -> (kernel name substring, units per profiled dispatch, number of trailing dispatches to keep, unit name)
CONFIGS = {
    "hmc|normal32|65536|fd_sparse|L16": ("k_hmc_sep_steps", 25, 4, "transition"),
    "hmc|normal32|8192|fd_sparse|L16": ("k_hmc_sep_steps", 25, 4, "transition"),
    "hmc|normal32|65536|fd_dense|L16": ("k_hmc_sep_steps", 25, 3, "transition"),
    "hmc|c3|65536|fd_sparse|L16": ("k_hmc_lin_steps", 1, 2, "transition"),
    "hmc|c3|8192|fd_sparse|L16": ("k_hmc_lin_steps", 1, 3, "transition"),
    "mh|refmodel20|65536": ("k_mh_mw", 100, 6, "chain step of every chain"),   # k_mh_mw_jit_steps since round 4 (every statement compiled at run time)
    "mh|refmodel20|8192": ("k_mh_mw", 100, 6, "chain step of every chain"),
    "mh|c5|262144": ("k_mh_mw", 100, 2, "chain step of every chain"),       # k_mh_mw_jit_steps since the general records are compiled at run time
    # the pipelined multi-wave MH kernel around statements compiled at run time (k_mh_mw_jit_steps): general stream records, no stream at all
    "mh|zoo:hier_scale|65536": ("_steps", 100, 2, "chain step of every chain"),
    "mh|zoo:alldists|65536": ("_steps", 100, 2, "chain step of every chain"),
    "mh|zoo:logistic100|65536": ("_steps", 100, 2, "chain step of every chain"),
    "smc|c4|1048576": ("", 1, 0, "run"),
    # programs without a record stream: compiled at run time (k_hmc_jit_steps); with FG_JIT=0 in the environment the interpreter kernel
    "hmc|zoo:alldists|65536|fd_sparse|L16": ("_steps", 5, 2, "transition"),
    "hmc|zoo:alldists|8192|fd_sparse|L16": ("_steps", 5, 2, "transition"),
    "hmc|zoo:hier_logsigma|65536|fd_sparse|L16": ("_steps", 5, 2, "transition"),
    "hmc|zoo:poisson_glm|65536|fd_sparse|L16": ("_steps", 5, 2, "transition"),
    "hmc|zoo:logistic100|65536|fd_sparse|L16": ("_steps", 5, 2, "transition"),
    # gradient-stream programs on the compiled unit (task code per wave, one barrier per gradient): the engine's choice since the end of round 4
    "hmc|zoo:refmodel8|65536|fd_sparse|L16": ("_steps", 5, 2, "transition"),
    "hmc|zoo:refmodel8|8192|fd_sparse|L16": ("_steps", 5, 2, "transition"),
    "hmc|zoo:hier|65536|fd_sparse|L16": ("_steps", 5, 2, "transition"),
}


def main(key):
    parts = key.split("|")
    if parts[0] == "hmc" and parts[1] == "normal32":
        C, mode = int(parts[2]), {"fd_sparse": E.GRAD_FD_SPARSE, "fd_dense": E.GRAD_FD_DENSE}[parts[3]]
        cp = E.compile_model(W.normal_sites(32))
        eng = E.Engine(cp, C, seed=1)
        n = CONFIGS[key][2]
        d = eng.device_alloc(25 * cp.d * C * 8)
        eng.hmc_init(E.hmc_config(grad_mode=mode), 25 if mode == E.GRAD_FD_SPARSE else 0)
        eng.hmc_step(25 if mode == E.GRAD_FD_SPARSE else 10)
        for _ in range(n):
            eng.hmc_step(25, d)                      # sampling launches: draw rows written, as in the bench's timed region
        eng.synchronize()
        print(key, eng.hmc_last_kernel())
    elif parts[0] == "hmc" and parts[1].startswith("zoo:"):
        from tests.models import ZOO
        C = int(parts[2])
        cp = E.compile_model(W.logistic_regression(*W.classification_data(100)[:2]) if parts[1] == "zoo:logistic100" else ZOO[parts[1][4:]]())
        eng = E.Engine(cp, C, seed=2)
        eng.hmc_init(E.hmc_config(grad_mode=E.GRAD_FD_SPARSE, n_leapfrog=16), 10)
        eng.hmc_step(10)                                 # adaptation
        for _ in range(CONFIGS[key][2]):
            eng.hmc_step(5)
        eng.synchronize()
        print(key, eng.hmc_last_kernel())
    elif parts[0] == "mh" and parts[1].startswith("zoo:"):
        from tests.models import ZOO
        C = int(parts[2])
        cp = E.compile_model(W.logistic_regression(*W.classification_data(100)[:2]) if parts[1] == "zoo:logistic100" else ZOO[parts[1][4:]]())
        eng = E.Engine(cp, C, seed=2)
        eng.mh_init(200)
        for _ in range(4):                           # 2 adapting, then the 2 sampling launches that are kept
            eng.mh_step(100)
        eng.synchronize()
        print(key, eng.mh_last_kernel())
    elif parts[0] == "hmc" and parts[1] == "c3":
        C = int(parts[2])
        X, y, _ = W.ridge_data(1024, 32)
        cp = E.compile_model(W.ridge_regression(X, y))
        eng = E.Engine(cp, C, seed=3)
        eng.hmc_init(E.hmc_config(init_step_size=0.004), 0)
        for _ in range(1 + CONFIGS[key][2]):
            eng.hmc_step(1)
        eng.synchronize()
        print(key, eng.hmc_last_kernel())
    elif key in ("mh|refmodel20|65536", "mh|refmodel20|8192"):
        eng = E.Engine(E.compile_model(W.reference_model(20)), int(parts[2]), seed=1)
        eng.mh_init(200)
        eng.mh_step(100)
        eng.mh_init(200)
        for _ in range(6):                           # 2 adapting + 4 sampling launches of 100 steps: the bench leg's mix
            eng.mh_step(100)
        eng.synchronize()
    elif key == "mh|c5|262144":
        data, _ = W.mixture_data(64)
        eng = E.Engine(E.compile_model(W.mixture(data)), 262144, seed=1)
        eng.mh_init(200)
        for _ in range(4):                           # 2 adapting, then the 2 sampling launches that are kept
            eng.mh_step(100)
        eng.synchronize()
    elif key == "smc|c4|1048576":
        eng = E.Engine(E.compile_model(W.smc_normal()), 1 << 20, seed=42)
        for _ in range(3):                           # the collector keeps the last run (from its k_smc_init on)
            r = eng.smc_run(rejuvenation_steps=3, download=False)
        print(key, len(r["betas"]), "tempering steps")
    else:
        raise SystemExit("unknown key " + key)


if __name__ == "__main__":
    main(sys.argv[1])
