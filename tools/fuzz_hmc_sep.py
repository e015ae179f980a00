"""Randomised cross-check of the register-resident HMC kernel against the gradient-stream kernel: random independent-sites
models (0..3 observations per site, sigmas that are powers of two / need the constant division / need IEEE division, constant
statements), random chain counts, leapfrog lengths, waves per tile, with and without mass adaptation -- draws, final state, step
sizes and divergence counts must agree bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fugue_amd as F
from fugue_amd import engine as E

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_models = int(sys.argv[2]) if len(sys.argv) > 2 else 24
SIG = [0.25, 0.5, 1.0, 2.0, 0.3, 1.7, 3.0, 1e-3, 123.456, 2.0 ** -30, 1.9999999999999998]


def random_model():
    d = int(rng.integers(1, 40))
    stm = []
    if rng.random() < 0.4:                                     # one record shape, powers of two, no constant statement: the dense mode's register-resident form
        nobs = int(rng.integers(0, 2))
        d = int(rng.integers(1, 70))
        for i in range(d):
            mu0, s0 = (0.0, 1.0) if rng.random() < 0.3 else (float(rng.normal()), float(rng.choice(SIG[:4])))
            stm.append((mu0, s0, [(float(rng.normal(scale=2.0)), float(rng.choice(SIG[:4])), bool(rng.integers(0, 2))) for _ in range(nobs)]))
        return build(stm, 0), d, True
    for i in range(d):
        mu0, s0 = float(rng.normal()), float(rng.choice(SIG[:7]))
        if rng.random() < 0.35: mu0, s0 = 0.0, 1.0           # the standard-normal own record has its own instance
        obs = [(float(rng.normal(scale=2.0)), float(rng.choice(SIG)), bool(rng.integers(0, 2))) for _ in range(int(rng.integers(0, 4)))]
        stm.append((mu0, s0, obs))
    n_const = int(rng.integers(0, 3))
    return build(stm, n_const), d, False


def build(stm, n_const):
    def model():
        m = F.pure(None)
        for i, (mu0, s0, obs) in enumerate(stm):
            def site(i=i, mu0=mu0, s0=s0, obs=obs):
                return F.sample(F.addr("x", i), F.Normal(mu0, s0)).bind(lambda x: F.sequence_vec(
                    [F.observe(F.addr("y", 10 * i + j), F.Normal(x, s) if own else F.Normal(yv, s), yv if own else x) for j, (yv, s, own) in enumerate(obs)]))
            m = m.bind(lambda _, site=site: site())
        for k in range(n_const):
            m = m.bind(lambda _, k=k: F.observe(F.addr("c", k), F.Normal(0.3 * k, 1.5), 0.1))
        return m
    return model


bad = 0
for it in range(n_models):
    model, d, shaped = random_model()
    try:
        cp = E.compile_model(model)
    except Exception as ex:
        print("model", it, "not compilable:", ex); continue
    C, L = int(rng.integers(1, 300)), int(rng.integers(1, 20))
    nw, ns = int(rng.integers(0, 30)), int(rng.integers(1, 25))
    mass = bool(rng.integers(0, 2)) and nw >= 20
    eps0 = None
    if rng.integers(0, 3) == 0:                                    # a pinned, far too large step: trajectories blow up (the checked re-run path)
        eps0, nw, mass = float(rng.choice([2.0, 50.0, 1e6, 1e160])), 0, False
    W = int(rng.choice([0, 1, 2, 4, 8, 16]))
    mode = [E.GRAD_FD_SPARSE, E.GRAD_FD_SPARSE, E.GRAD_FD_DENSE, E.GRAD_ANALYTIC][int(rng.integers(0, 4))]
    if shaped and rng.random() < 0.8: mode = E.GRAD_FD_DENSE
    out = []
    for sep in (1, 0):
        os.environ["FG_HMC_SEP"] = str(sep)
        os.environ["FG_HMC_WAVES"] = str(W if sep else 0)
        eng = E.Engine(cp, C, seed=100 + it)
        buf = eng.device_alloc(max(1, ns * cp.d * C) * 8)
        st = eng.hmc_run(E.hmc_config(n_leapfrog=L, adapt_mass=mass, init_step_size=eps0, grad_mode=mode), ns, nw, buf)
        out.append((eng.download(buf, (ns, cp.d, C), dtype=np.int64), eng.get_values(), eng.hmc_step_sizes(), eng.hmc_log_joint(), int(st.n_divergent), eng.hmc_mass()))
        if sep: kern = eng.hmc_last_kernel()
        eng.device_free(buf); eng.close()
    ok = all(np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True) for a, b in zip(out[0], out[1]))
    bad += 0 if ok else 1
    n_sep = E.lib().fg_program_stream_records(cp.h, 3)
    print(f"model {it:3d}: d={d:2d} records={cp.stream_records} sep={n_sep:3d} C={C:3d} L={L:2d} warm={nw:2d} n={ns:2d} mass={int(mass)} W={W:2d} mode={mode} eps0={eps0} div={out[0][4]:4d} [{kern[16:56]}] -> {'identical' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
