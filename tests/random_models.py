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


def random_expression_program(seed: int) -> M.Program:
    """Programs whose parameters are random EXPRESSIONS of the sites -- every opcode of the interpreter (arithmetic, exp / ln / sqrt /
    abs / floor / sin / cos / tanh / pow / min / max / clamp / select, linear predictors long enough to fuse), every distribution family
    as prior or likelihood, plates of isomorphic observe statements, a factor statement: what the run-time compiler (fg_jit.cpp) has to
    reproduce instruction for instruction.  Values may leave a family's support on some chains (-inf weights, NaN): both paths must
    agree on those as well."""
    rng = np.random.default_rng(1000 + seed)
    P = M.Program()
    n_real = int(rng.integers(2, 6))
    xs = [P.sample(M.addr("x", i), M.Normal(float(rng.normal(scale=0.5)), float(rng.choice(SIG)))) for i in range(n_real)]
    pos = [P.sample(M.addr("s", 0), M.Gamma(3.0, 2.0))]
    if rng.random() < 0.5:
        pos.append(P.sample(M.addr("s", 1), M.LogNormal(0.1, 0.5)))
    unit = P.sample(M.addr("u"), M.Beta(2.0, 2.5)) if rng.random() < 0.6 else None
    disc = P.sample(M.addr("z"), M.Categorical([0.2, 0.5, 0.3])) if rng.random() < 0.5 else None

    def leaf():
        k = rng.random()
        if k < 0.55: return xs[int(rng.integers(n_real))]
        if k < 0.75: return pos[int(rng.integers(len(pos)))]
        if k < 0.85 and unit is not None: return unit
        return M.as_expr(float(np.round(rng.normal(), 2)))

    def expr(depth):
        if depth == 0 or rng.random() < 0.25:
            return leaf()
        op = str(rng.choice(["add", "sub", "mul", "div", "neg", "exp", "ln", "sqrt", "abs", "floor", "sin", "cos", "tanh", "pow", "min", "max", "clamp", "lin"]))
        a = expr(depth - 1)
        if op == "add": return a + expr(depth - 1)
        if op == "sub": return a - expr(depth - 1)
        if op == "mul": return a * expr(depth - 1)
        if op == "div": return a / (2.0 + M.fabs(expr(depth - 1)))
        if op == "neg": return -a
        if op == "exp": return M.exp(M.clamp(a, -3.0, 3.0))
        if op == "ln": return M.ln(1.0 + M.fabs(a))
        if op == "sqrt": return M.sqrt(M.fabs(a))
        if op == "abs": return M.fabs(a)
        if op == "floor": return M.floor(a)
        if op == "sin": return M.sin(a)
        if op == "cos": return M.cos(a)
        if op == "tanh": return M.tanh(a)
        if op == "pow": return M.powf(1.0 + M.fabs(a), float(rng.choice([0.5, 1.5, 2.0])))
        if op == "min": return M.fmin(a, expr(depth - 1))
        if op == "max": return M.fmax(a, expr(depth - 1))
        if op == "clamp": return M.clamp(a, -1.5, 2.0)
        lp = M.as_expr(float(np.round(rng.normal(), 2)))               # a linear predictor of five or six terms (fuses into one DOT instruction)
        for _ in range(int(rng.integers(5, 7))):
            lp = lp + xs[int(rng.integers(n_real))] * float(np.round(rng.normal(), 3))
        return lp

    def positive(depth): return 0.05 + M.fabs(expr(depth))
    def prob(depth): return M.clamp(1.0 / (1.0 + M.exp(-M.clamp(expr(depth), -8.0, 8.0))), 1e-6, 1.0 - 1e-6)

    n_stmt = int(rng.integers(3, 9))
    for j in range(n_stmt):
        fam = str(rng.choice(["normal", "normal", "poisson", "bern", "gamma", "student", "laplace", "cauchy", "lognormal", "expo", "beta", "weibull", "binomial", "select", "plate"]))
        y = float(np.round(rng.normal(), 2))
        if fam == "normal": P.observe(M.addr("o", j), M.Normal(expr(2), positive(1)), y)
        elif fam == "poisson": P.observe(M.addr("o", j), M.Poisson(positive(2)), int(rng.integers(0, 6)))
        elif fam == "bern": P.observe(M.addr("o", j), M.Bernoulli(prob(2)), bool(rng.integers(0, 2)))
        elif fam == "gamma": P.observe(M.addr("o", j), M.Gamma(positive(1), positive(1)), abs(y) + 0.1)
        elif fam == "student": P.observe(M.addr("o", j), M.StudentT(2.0 + positive(1), expr(1), positive(1)), y)
        elif fam == "laplace": P.observe(M.addr("o", j), M.Laplace(expr(2), 0.8), y)
        elif fam == "cauchy": P.observe(M.addr("o", j), M.Cauchy(expr(1), positive(1)), y)
        elif fam == "lognormal": P.observe(M.addr("o", j), M.LogNormal(expr(1), 0.5), abs(y) + 0.1)
        elif fam == "expo": P.observe(M.addr("o", j), M.Exponential(positive(2)), abs(y))
        elif fam == "beta": P.observe(M.addr("o", j), M.Beta(positive(1), positive(1)), float(rng.uniform(0.05, 0.95)))
        elif fam == "weibull": P.observe(M.addr("o", j), M.Weibull(positive(1), positive(1)), abs(y) + 0.1)
        elif fam == "binomial": P.observe(M.addr("o", j), M.Binomial(8, prob(1)), int(rng.integers(0, 9)))
        elif fam == "select" and disc is not None: P.observe(M.addr("o", j), M.Normal(M.select(disc, [xs[0], expr(1), 0.5]), 1.0), y)
        else:                                                                # a plate: the same statement shape, different constants (rolls into a loop)
            w = [float(np.round(rng.normal(), 3)) for _ in range(3)]
            for r in range(int(rng.integers(4, 9))):
                eta = xs[0] * float(np.round(rng.normal(), 3)) + xs[1] * float(np.round(rng.normal(), 3)) + w[0]
                if j % 2: P.observe(M.addr("p", 100 * j + r), M.Poisson(M.exp(M.clamp(eta, -4.0, 4.0))), int(rng.integers(0, 5)))
                else: P.observe(M.addr("p", 100 * j + r), M.Normal(eta, pos[0]), float(np.round(rng.normal(), 2)))
    if rng.random() < 0.5:
        P.factor(-0.1 * xs[0] * xs[0] + 0.05 * xs[-1])
    return P
