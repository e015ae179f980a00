"""Model zoo shared by the CPU and GPU tests (descriptions only)."""
import numpy as np

from fugue_amd import model as M
from fugue_amd import workloads as W


def all_dists_model() -> M.Program:
    """Every distribution once with site-dependent parameters (non-hoisted path) and once with
    constant parameters (hoisted path), plus expressions, factor and select."""
    P = M.Program()
    a = P.sample(M.addr("a"), M.Normal(0.0, 1.0))
    s = P.sample(M.addr("s"), M.Gamma(2.0, 1.5))
    b = P.sample(M.addr("b"), M.Beta(2.0, 3.0))
    u = P.sample(M.addr("u"), M.Uniform(-2.0, 2.0))
    ln = P.sample(M.addr("ln"), M.LogNormal(a, s))
    e = P.sample(M.addr("e"), M.Exponential(s))
    be = P.sample(M.addr("be"), M.Bernoulli(b))
    cat = P.sample(M.addr("cat"), M.Categorical([b, 1.0 - b]))
    cat2 = P.sample(M.addr("cat", 2), M.Categorical([0.2, 0.3, 0.5]))
    bi = P.sample(M.addr("bi"), M.Binomial(10, b))
    po = P.sample(M.addr("po"), M.Poisson(s * 2.0))
    t = P.sample(M.addr("t"), M.StudentT(3.0 + s, a, s))
    c = P.sample(M.addr("c"), M.Cauchy(a, s))
    la = P.sample(M.addr("la"), M.Laplace(a, s))
    w = P.sample(M.addr("w"), M.Weibull(1.0 + s, 1.0 + e))
    ch = P.sample(M.addr("ch"), M.ChiSquared(1.0 + s))
    ig = P.sample(M.addr("ig"), M.InverseGamma(2.0 + s, 1.0 + s))
    du = P.sample(M.addr("du"), M.DiscreteUniform(-2, 5))
    # hoisted observes (constant parameters)
    P.observe(M.addr("o", 0), M.Normal(0.3, 0.5), a)
    P.observe(M.addr("o", 1), M.Normal(0.3, 0.7), 0.25)
    P.observe(M.addr("o", 2), M.LogNormal(0.1, 2.0), ln)
    P.observe(M.addr("o", 3), M.Exponential(1.3), e)
    P.observe(M.addr("o", 4), M.Bernoulli(0.3), be)
    P.observe(M.addr("o", 5), M.Beta(2.5, 1.5), b)
    P.observe(M.addr("o", 6), M.Gamma(3.0, 2.0), s)
    P.observe(M.addr("o", 7), M.Binomial(12, 0.4), bi)
    P.observe(M.addr("o", 8), M.Poisson(3.5), po)
    P.observe(M.addr("o", 9), M.StudentT(4.0, 0.5, 2.0), t)
    P.observe(M.addr("o", 10), M.Cauchy(0.5, 0.25), c)
    P.observe(M.addr("o", 11), M.Laplace(-0.5, 4.0), la)
    P.observe(M.addr("o", 12), M.Weibull(1.5, 2.0), w)
    P.observe(M.addr("o", 13), M.ChiSquared(4.0), ch)
    P.observe(M.addr("o", 14), M.InverseGamma(3.0, 2.0), ig)
    P.observe(M.addr("o", 15), M.DiscreteUniform(-3, 8), du)
    P.observe(M.addr("o", 16), M.Uniform(-3.0, 3.0), u)
    P.observe(M.addr("o", 17), M.Categorical([0.1, 0.2, 0.3, 0.4]), cat2)
    # expressions
    P.observe(M.addr("o", 18), M.Normal(M.select(cat2, [a, u, b]), 1.0), 0.3)
    P.observe(M.addr("o", 19), M.Normal(M.exp(-M.fabs(a)) + M.sqrt(s) * M.tanh(u) - M.ln(1.0 + e) / (2.0 + M.cos(c)),
                                        M.fmax(0.1, M.fmin(s, 3.0)) ** 1.5), 0.7)
    P.observe(M.addr("o", 20), M.Normal((a * 0.5 + u * 0.25 + b * 2.0).clamp(-1.0, 1.0), 1.0 + M.floor(s) + M.sin(a) ** 2.0), -0.2)
    P.observe(M.addr("o", 21), M.Poisson(M.exp(0.3 * a) + cat), 2)
    P.factor(-0.5 * a * a + 0.1 * u)
    return P


def f64_values_for(orc_model, rng: np.random.Generator, n_chains: int) -> np.ndarray:
    """Random but in-support cells [S][C] for `all_dists_model`-like programs: draw from the prior."""
    cells = np.zeros((orc_model.S, n_chains), dtype=np.int64)
    for c in range(n_chains):
        v, _, _ = orc_model.run_prior(int(rng.integers(1 << 30)), c)
        cells[:, c] = v
    return cells


ZOO = {
    "readme": lambda: W.readme_normal(),
    "normal32": lambda: W.normal_sites(32),
    "coin": lambda: W.coin_flip(),
    "refmodel8": lambda: W.reference_model(8),
    "ridge": lambda: W.ridge_regression(*W.ridge_data(24, 4)[:2]),
    "mixture": lambda: W.mixture(W.mixture_data(10)[0]),
    "alldists": all_dists_model,
}


def hier_normal(groups: int = 5, per_group: int = 3) -> M.Program:
    """Hierarchical Normal with known scales: every force term is a Normal whose x and mu are sites or
    constants, so the fused gradient stream applies, with coordinates that read EACH OTHER (mu <-> m#g)
    and very different record counts per coordinate (mu has 1 + groups, m#g has 2 + per_group)."""
    P = M.Program()
    mu = P.sample(M.addr("mu"), M.Normal(0.5, 2.0))
    rng = np.random.default_rng(3)
    for g in range(groups):
        m = P.sample(M.addr("m", g), M.Normal(mu, 0.7))
        P.observe(M.addr("anchor", g), M.Normal(m, 3.0), 0.25 * g)          # mu = site, sigma not a power of two
        for j in range(per_group):
            P.observe(M.addr("y", g * per_group + j), M.Normal(m, 0.5 + 0.25 * j), float(rng.normal(0.3 * g, 1.0)))
    return P


ZOO["hier"] = hier_normal
ZOO["ridge7"] = lambda: W.ridge_regression(*W.ridge_data(10, 7)[:2], sigma=0.8)     # FG_OP_DOT with a 4 + 3 term split, sigma not 2^k


def hier_scale() -> M.Program:
    """Leaf-operand statements of many families (the general FG_G_GEN stream records): unknown scales as sites, a
    Normal whose sigma is a site, Student-t / Gamma / Exponential / Beta / LogNormal priors, Bernoulli and Poisson
    likelihoods whose parameter is a site."""
    P = M.Program()
    tau = P.sample(M.addr("tau"), M.Gamma(2.0, 2.0))
    s = P.sample(M.addr("s"), M.Exponential(1.5))
    mu = P.sample(M.addr("mu"), M.Normal(0.3, 5.0))
    b = P.sample(M.addr("b"), M.Beta(2.0, 3.0))
    ln = P.sample(M.addr("ln"), M.LogNormal(0.1, 0.7))
    for g in range(3):
        x = P.sample(M.addr("x", g), M.Normal(mu, tau))                  # sigma is a site
        P.observe(M.addr("y", g), M.Normal(x, 0.5), 0.4 * g - 0.2)
        P.observe(M.addr("t", g), M.StudentT(4.0, x, s), 0.1 * g + 0.3)  # scale is a site
    for i, f in enumerate([1, 0, 1, 1]):
        P.observe(M.addr("flip", i), M.Bernoulli(b), float(f))
    P.observe(M.addr("count"), M.Poisson(ln), 3)
    P.observe(M.addr("c"), M.Cauchy(mu, 2.0), 0.7)
    return P


ZOO["hier_scale"] = hier_scale
# dense regressions with 8 / 16 / 32 coefficients: the observation-major gradient kernel (fg_hmc_lin.hip), sigma 2^k and not
ZOO["ridge8"] = lambda: W.ridge_regression(*W.ridge_data(20, 8)[:2])
ZOO["ridge16"] = lambda: W.ridge_regression(*W.ridge_data(12, 16)[:2], sigma=0.8)
ZOO["ridge32"] = lambda: W.ridge_regression(*W.ridge_data(16, 32)[:2])
ZOO["ridge24"] = lambda: W.ridge_regression(*W.ridge_data(20, 24)[:2], sigma=0.8)    # dense regressions between the built sizes: padded term positions
ZOO["ridge12"] = lambda: W.ridge_regression(*W.ridge_data(14, 12)[:2])
ZOO["ridge5"] = lambda: W.ridge_regression(*W.ridge_data(9, 5)[:2])
ZOO["ridge64"] = lambda: W.ridge_regression(*W.ridge_data(12, 64)[:2])               # 64 term positions: sixteen waves, q read from LDS
ZOO["ridge40"] = lambda: W.ridge_regression(*W.ridge_data(10, 40)[:2], sigma=0.8)


def linreg_forms() -> M.Program:
    """Linear predictors written the ways users write them: alpha + beta*x, beta*x + alpha, gamma*u + beta,
    alpha + gamma -- all FG_G_LIN records (LOAD slot, MUL c, MAC, ADD slot)."""
    P = M.Program()
    alpha = P.sample(M.addr("alpha"), M.Normal(0.0, 2.0))
    beta = P.sample(M.addr("beta"), M.Normal(0.0, 2.0))
    gamma = P.sample(M.addr("gamma"), M.Normal(0.0, 1.0))
    xs = [-1.5, -0.5, 0.25, 1.0, 2.0]
    for i, x in enumerate(xs):
        P.observe(M.addr("y", i), M.Normal(alpha + beta * x, 0.8), 0.4 + 0.9 * x)
        P.observe(M.addr("z", i), M.Normal(beta * x + alpha, 0.5), 0.5 + 0.8 * x)
        P.observe(M.addr("w", i), M.Normal(gamma * (0.3 * i - 0.4) + beta, 1.0), 0.1 * i)
    P.observe(M.addr("v"), M.Normal(alpha + gamma, 1.0), 0.6)
    return P


ZOO["linreg"] = linreg_forms


def indep_mixed() -> M.Program:
    """Independent sites with every shape the register-resident trajectory kernel handles: 1 to 4 force terms per
    coordinate, sigma a power of two / not / outside the exact-reciprocal range, the coordinate as x or as mu, a
    constant-only statement (no coordinate), an odd number of coordinates."""
    P = M.Program()
    rng = np.random.default_rng(17)
    sig = [1.0, 0.7, 2.0, 0.3, 1.5, 0.25, 1e-3]
    P.observe(M.addr("const_obs"), M.Normal(0.2, 1.3), 0.9)                  # reads no coordinate
    for i in range(7):
        x = P.sample(M.addr("x", i), M.Normal(0.1 * i - 0.2, sig[i]))
        for j in range(i % 4):                                               # 0..3 observes -> 1..4 records
            if j % 2 == 0:
                P.observe(M.addr(f"y{j}", i), M.Normal(x, 0.5 + 0.35 * j), float(rng.normal(0.1 * i, 0.5)))
            else:
                P.observe(M.addr(f"y{j}", i), M.Normal(0.3 * j, sig[(i + j) % 6]), x)      # the coordinate is the VALUE of the observe
    return P


ZOO["indep_mixed"] = indep_mixed


def indep_uniform5() -> M.Program:
    """Independent sites that all have ONE record shape with power-of-two sigmas (what the half-tile layout of the register-resident
    kernel needs): an odd number of coordinates, a non-standard prior and two observes each."""
    P = M.Program()
    rng = np.random.default_rng(23)
    for i in range(5):
        x = P.sample(M.addr("x", i), M.Normal(0.3 - 0.1 * i, 2.0))
        P.observe(M.addr("ya", i), M.Normal(x, 0.5), float(rng.normal(0.2 * i, 0.5)))
        P.observe(M.addr("yb", i), M.Normal(x, 0.25), float(rng.normal(0.2 * i, 0.25)))
    return P


ZOO["indep_uniform5"] = indep_uniform5


def poisson_glm(n: int = 12) -> M.Program:
    """A log-link count regression: the Poisson rate is exp(linear predictor) -- an expression parameter, so every coordinate's
    finite difference runs through the interpreter (no record stream)."""
    rng = np.random.default_rng(12)
    P = M.Program()
    b = [P.sample(M.addr("b", j), M.Normal(0.0, 1.0 if j else 2.0)) for j in range(4)]
    X = rng.normal(0.0, 0.6, (n, 3))
    for i in range(n):
        eta = b[0] + b[1] * float(X[i, 0]) + b[2] * float(X[i, 1]) + b[3] * float(X[i, 2])
        P.observe(M.addr("y", i), M.Poisson(M.exp(eta)), int(rng.poisson(np.exp(0.4 + 0.5 * X[i, 0] - 0.3 * X[i, 1]))))
    return P


def hier_logsigma(groups: int = 5) -> M.Program:
    """A hierarchical model in the usual unconstrained form: the group scale is exp(log_sigma) and the observation scale a
    sqrt of a site -- expression parameters on both levels."""
    rng = np.random.default_rng(13)
    P = M.Program()
    mu = P.sample(M.addr("mu"), M.Normal(0.0, 5.0))
    ls = P.sample(M.addr("log_sigma"), M.Normal(0.0, 1.0))
    v = P.sample(M.addr("v"), M.Gamma(3.0, 2.0))
    for g in range(groups):
        x = P.sample(M.addr("x", g), M.Normal(mu, M.exp(ls)))
        for j in range(2):
            P.observe(M.addr("y", 2 * g + j), M.Normal(x, M.sqrt(v)), float(rng.normal(0.5 * g, 1.0)))
    return P


ZOO["logistic"] = lambda: W.logistic_regression(*W.classification_data(14)[:2])     # the reference's classification example, small
ZOO["poisson_glm"] = poisson_glm
ZOO["hier_logsigma"] = hier_logsigma


# random programs (tests/random_models.py): every ZOO-wide test (site tables, log-joint, prior draws vs the oracle) covers them too
from tests.random_models import random_program  # noqa: E402

for _k in range(6):
    ZOO[f"rand{_k}"] = (lambda k: (lambda: random_program(2000 + k)))(_k)
