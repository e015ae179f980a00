"""Randomised cross-check of the multi-wave MH kernel against the one-wave kernel: random models out of chained Normal sites,
Gamma / Beta / Exponential scale and rate sites, Categorical sites with random tables (zeros included) selecting among sites
and constants, Poisson and Bernoulli sites, random chain counts, warmup / sampling lengths and waves per tile -- recorded draws,
final state, adapted scales, log-weights and accept counts must be identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fugue_amd as F
from fugue_amd import model as M
from fugue_amd import engine as E

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_models = int(sys.argv[2]) if len(sys.argv) > 2 else 30
SIG = [0.25, 0.5, 1.0, 2.0, 0.3, 1.7]


def random_model():
    spec = []
    n_f = int(rng.integers(1, 8))
    for i in range(n_f):
        spec.append(("chain", float(rng.normal()), float(rng.choice(SIG)), float(rng.choice([0.0, 0.5, 1.0, -0.7])), int(rng.integers(0, 3))))
    extras = [str(rng.choice(["gamma", "beta", "expo", "cat", "cat", "poisson", "bern", "none"])) for _ in range(int(rng.integers(0, 5)))]
    cats = [(int(rng.integers(2, 7)), int(rng.integers(1, 4))) for e in extras if e == "cat"]
    tables = []
    for K, _ in cats:
        p = rng.random(K) * (rng.random(K) > 0.2)
        if p.sum() == 0: p[0] = 1.0
        tables.append(list(p / p.sum()))
    obs_y = [float(rng.normal(scale=2.0)) for _ in range(64)]

    def model():
        P = F.Program()
        xs = []
        prev = None
        for i, (_, m0, s0, a, nobs) in enumerate(spec):
            mu = m0 if prev is None else m0 + a * prev
            x = P.sample(F.addr("x", i), F.Normal(mu, s0))
            for j in range(nobs):
                P.observe(F.addr("y", 10 * i + j), F.Normal(x, 0.5 + 0.25 * j), obs_y[(3 * i + j) % 64])
            xs.append(x); prev = x
        ci = 0
        for e_i, e in enumerate(extras):
            if e == "gamma":
                g = P.sample(F.addr("g", e_i), F.Gamma(3.0, 2.0)); P.observe(F.addr("yg", e_i), F.Normal(xs[0], g), obs_y[e_i])
            elif e == "beta":
                b = P.sample(F.addr("b", e_i), F.Beta(2.0, 3.0)); P.observe(F.addr("yb", e_i), F.Bernoulli(b), bool(e_i & 1))
            elif e == "expo":
                r = P.sample(F.addr("r", e_i), F.Exponential(1.5)); P.observe(F.addr("yr", e_i), F.Poisson(r), int(e_i + 1))
            elif e == "cat":
                K, n_o = cats[ci]
                z = P.sample(F.addr("z", e_i), F.Categorical(tables[ci]))
                opts = [xs[k % len(xs)] if (k + e_i) % 2 == 0 else float(k) - 1.0 for k in range(K)]
                for j in range(n_o):
                    P.observe(F.addr("yz", 10 * e_i + j), F.Normal(M.select(z, opts), 0.7), obs_y[(7 * e_i + j) % 64])
                ci += 1
            elif e == "poisson":
                P.sample(F.addr("k", e_i), F.Poisson(3.0))
            elif e == "bern":
                P.sample(F.addr("flag", e_i), F.Bernoulli(0.3))
        return P
    return model


bad = 0
for it in range(n_models):
    model = random_model()
    try:
        cp = E.compile_model(model())
    except Exception as ex:
        print("model", it, "not compilable:", repr(ex)[:100]); continue
    C, nw, ns = int(rng.integers(1, 300)), int(rng.integers(0, 60)), int(rng.integers(1, 40))
    W = int(rng.choice([0, 2, 4, 8, 16]))
    out = []
    for mw in (1, 0):
        os.environ["FG_MH_MW"] = str(mw)
        os.environ["FG_HMC_WAVES"] = str(W if mw else 0)
        eng = E.Engine(cp, C, seed=500 + it, chain_offset=it)
        buf = eng.device_alloc(max(1, ns * cp.S * C) * 8)
        st = eng.mh_run(ns, nw, None, list(range(cp.S)), buf)
        out.append((eng.download(buf, (ns, cp.S, C), dtype=np.int64), eng.get_values(), eng.mh_scales(), eng.mh_log_weight(), st.accept_rate))
        eng.device_free(buf); eng.close()
    ok = all(np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True) for a, b in zip(out[0], out[1]))
    bad += 0 if ok else 1
    print(f"model {it:3d}: S={cp.S:2d} O={cp.O:2d} records={cp.stream_records} C={C:3d} warm={nw:2d} n={ns:2d} W={W:2d} accept={out[0][4]:.3f} -> {'identical' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
