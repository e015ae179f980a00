"""Random dense regressions on k_hmc_lin_steps against the gradient stream (FG_HMC_LIN=0), bit for bit: coefficient counts 2 ... 64,
1 ... 60 observations, sigma a power of two or not, prior scale 1 or not, mass adaptation on / off, full and half tiles, every waves-per-tile
layout the host accepts; every third model starts a few chains far out (|beta| ~ 1e150 ... 1e200: non-finite forces, the unfused re-run of the
eight-coordinate instances).  usage: python tools/fuzz_hmc_lin.py [seed] [n_models]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FG_JIT"] = "0"
import numpy as np
from fugue_amd import engine as E, workloads as W

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_models = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rs = np.random.RandomState(seed)
bad = 0
for m in range(n_models):
    p = int(rs.choice([2, 3, 5, 7, 8, 9, 12, 15, 16, 17, 24, 31, 32, 33, 40, 48, 64]))
    n = int(rs.randint(1, 61))
    sigma = float(rs.choice([0.5, 2.0, 0.25, 0.7, 1.3]))
    lam = float(rs.choice([1.0, 4.0, 0.37]))
    X = rs.standard_normal((n, p)); y = X @ rs.standard_normal(p) + sigma * rs.standard_normal(n)
    cp = E.compile_model(W.ridge_regression(X, y, sigma=sigma, lam=lam))
    C = int(rs.choice([64, 96, 150, 257]))
    adapt = bool(rs.randint(2)); far = m % 3 == 2
    D = 8 if p <= 8 else 16 if p <= 16 else 32 if p <= 32 else 64
    layouts = [(0, 0, 0)] + [(1, w, h) for w in sorted({D // 8, D // 4, D // 2} - {0, 1}) for h in ((0, 1) if D < 64 else (0,)) if not (D == 64 and w == D // 2)]
    out, names = [], []
    for lin, w, half in layouts:
        os.environ["FG_HMC_LIN"] = str(lin); os.environ["FG_HMC_LIN_HALF"] = str(half)
        if w: os.environ["FG_HMC_WAVES"] = str(w)
        else: os.environ.pop("FG_HMC_WAVES", None)
        eng = E.Engine(cp, C, seed=100 + m, chain_offset=3)
        if far:
            eng.hmc_init(E.hmc_config(n_leapfrog=4, init_step_size=1e-3, adapt_mass=False), 0)
            cells = np.ascontiguousarray(eng.get_values()).view(np.float64).copy()
            r2 = np.random.RandomState(1000 + m)
            for c in range(0, C, 3):
                for j in r2.choice(cp.S, size=1 + c % 2, replace=False): cells[j, c] = r2.choice([-1.0, 1.0]) * 10.0 ** r2.uniform(150.0, 200.0)
            eng.set_values(cells.view(np.int64)); eng.hmc_step(3)
            st = eng.hmc_stats()
            out.append((eng.get_values(), eng.hmc_step_sizes(), eng.hmc_log_joint(), st.n_divergent))
        else:
            d = eng.device_alloc(6 * cp.d * C * 8)
            st = eng.hmc_run(E.hmc_config(n_leapfrog=5, adapt_mass=adapt), 6, 12, d)
            out.append((eng.download(d, (6, cp.d, C)), eng.hmc_step_sizes(), eng.hmc_log_joint(), eng.get_values(), st.n_divergent, eng.hmc_mass() if adapt else None))
            eng.device_free(d)
        names.append(eng.hmc_last_kernel()); eng.close()
    same = all(all((a is None and b is None) or np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True) for a, b in zip(out[0], o)) for o in out[1:])
    on_lin = all("k_hmc_lin_steps" in k for k in names[1:]) and "k_hmc_lin_steps" not in names[0]
    print(f"model {m:3d}: p={p:2d} n={n:2d} sigma={sigma} lam={lam} C={C} adapt={int(adapt)} far={int(far)} layouts={len(layouts) - 1} -> {'identical' if same and on_lin else 'MISMATCH ' + str(names)}", flush=True)
    bad += 0 if same and on_lin else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
