"""Random site programs for the randomised parity tests: chained Normal sites with a varying number of observations, scale /
rate / probability sites feeding Normal, Bernoulli and Poisson likelihoods, Categorical sites with random tables (zeros
included) selecting among sites and constants, free Poisson and Bernoulli sites."""
import numpy as np

from fugue_amd import model as M

SIG = [0.25, 0.5, 1.0, 2.0, 0.3, 1.7]


def random_program(seed: int) -> M.Program:
    rng = np.random.default_rng(seed)
    obs_y = [float(v) for v in rng.normal(scale=2.0, size=64)]
    P = M.Program()
    xs, prev = [], None
    for i in range(int(rng.integers(1, 8))):
        m0, s0, a = float(rng.normal()), float(rng.choice(SIG)), float(rng.choice([0.0, 0.5, 1.0, -0.7]))
        x = P.sample(M.addr("x", i), M.Normal(m0 if prev is None else m0 + a * prev, s0))
        for j in range(int(rng.integers(0, 3))):
            P.observe(M.addr("y", 10 * i + j), M.Normal(x, 0.5 + 0.25 * j), obs_y[(3 * i + j) % 64])
        xs.append(x)
        prev = x
    for e_i in range(int(rng.integers(0, 5))):
        e = str(rng.choice(["gamma", "beta", "expo", "cat", "cat", "poisson", "bern", "none"]))
        if e == "gamma":
            g = P.sample(M.addr("g", e_i), M.Gamma(3.0, 2.0))
            P.observe(M.addr("yg", e_i), M.Normal(xs[0], g), obs_y[e_i])
        elif e == "beta":
            b = P.sample(M.addr("b", e_i), M.Beta(2.0, 3.0))
            P.observe(M.addr("yb", e_i), M.Bernoulli(b), bool(e_i & 1))
        elif e == "expo":
            r = P.sample(M.addr("r", e_i), M.Exponential(1.5))
            P.observe(M.addr("yr", e_i), M.Poisson(r), int(e_i + 1))
        elif e == "cat":
            K = int(rng.integers(2, 7))
            p = rng.random(K) * (rng.random(K) > 0.2)
            if p.sum() == 0.0:
                p[0] = 1.0
            z = P.sample(M.addr("z", e_i), M.Categorical([float(v) for v in p / p.sum()]))
            opts = [xs[k % len(xs)] if (k + e_i) % 2 == 0 else float(k) - 1.0 for k in range(K)]
            for j in range(int(rng.integers(1, 4))):
                P.observe(M.addr("yz", 10 * e_i + j), M.Normal(M.select(z, opts), 0.7), obs_y[(7 * e_i + j) % 64])
        elif e == "poisson":
            P.sample(M.addr("k", e_i), M.Poisson(3.0))
        elif e == "bern":
            P.sample(M.addr("flag", e_i), M.Bernoulli(0.3))
    return P
