"""The BASELINE.json configurations as model builders (descriptions only -- `model.Program`).

Each returns a `Program`; closed-form posteriors are documented beside them.
"""
from __future__ import annotations

import numpy as np

from .model import (Bernoulli, Beta, Categorical, Gamma, Normal, Program, addr, observe, plate, pure, sample, select,
                    trace_model)


def coin_flip(data=(1, 0, 1, 1, 0, 1, 1, 0, 1, 1)) -> Program:
    """C1: /root/reference/examples/bayesian_coin_flip.rs:9-25,36-38.  Posterior Beta(9,5)."""
    def model():
        return sample(addr("coin_bias"), Beta(2.0, 2.0)).bind(
            lambda p: plate(range(len(data)), lambda i: observe(addr("flip", i), Bernoulli(p.clamp(1e-10, 1.0 - 1e-10)),
                                                                bool(data[i]))).map(lambda _: p))
    return trace_model(model)


def readme_normal(y: float = 1.2, sigma: float = 0.5) -> Program:
    """C2 (d=1): /root/reference/README.md:74-80.  Posterior N(0.96, 0.2) for y=1.2, sigma=0.5."""
    return trace_model(lambda: sample(addr("mu"), Normal(0.0, 1.0)).bind(
        lambda mu: observe(addr("y"), Normal(mu, sigma), y).map(lambda _: mu)))


def normal_sites(n: int = 32) -> Program:
    """C2 north-star model: x#i ~ N(0,1); y#i ~ N(x#i, 0.5) observed at 0.2*i - 1 (datum rule of
    /root/reference/benches/f_perf.rs:84).  Posterior x#i ~ N(0.8*(0.2 i - 1), 0.2), independent."""
    P = Program()
    for i in range(n):
        x = P.sample(addr("x", i), Normal(0.0, 1.0))
        P.observe(addr("y", i), Normal(x, 0.5), 0.2 * i - 1.0)
    return P


def normal_sites_truth(n: int = 32):
    """(site order, means, variances) of `normal_sites` in engine coordinate order."""
    names = sorted((addr("x", i) for i in range(n)), key=lambda s: s.encode())
    idx = [int(s.split("#")[1]) for s in names]
    return names, np.array([0.8 * (0.2 * i - 1.0) for i in idx]), np.full(n, 0.2)


def reference_model(n_sites: int = 20) -> Program:
    """/root/reference/benches/f_perf.rs:78-91: mu ~ N(0,1); x#i ~ N(mu,1); y#i ~ N(x#i,0.5) = 0.2 i - 1."""
    P = Program()
    mu = P.sample(addr("mu"), Normal(0.0, 1.0))
    for i in range(max(0, n_sites - 1)):
        x = P.sample(addr("x", i), Normal(mu, 1.0))
        P.observe(addr("y", i), Normal(x, 0.5), 0.2 * i - 1.0)
    return P


def ridge_regression(X: np.ndarray, y: np.ndarray, sigma: float = 0.5, lam: float = 1.0) -> Program:
    """C3: ridge form of /root/reference/examples/linear_regression.rs:396-424 generalised to p
    coefficients: beta#j ~ N(0, 1/sqrt(lam)); y#i ~ N(sum_j beta#j X[i][j], sigma).
    Posterior N(mu, Sigma), Sigma = (X'X/sigma^2 + lam I)^-1, mu = Sigma X'y / sigma^2."""
    n, p = X.shape
    P = Program()
    betas = [P.sample(addr("beta", j), Normal(0.0, 1.0 / np.sqrt(lam))) for j in range(p)]
    for i in range(n):
        mean = 0.0
        for j in range(p):                      # mean_i += beta_j * x_i[j], starting from 0.0
            mean = mean + betas[j] * float(X[i, j])
        P.observe(addr("y", i), Normal(mean, sigma), float(y[i]))
    return P


def logistic_regression(X: np.ndarray, labels: np.ndarray) -> Program:
    """/root/reference/examples/classification.rs:104-135 (run there with adaptive_mcmc_chain, :159): beta#j ~ N(0, 2);
    linear_pred = 0.0 + sum_j beta#j x_ij in feature order; prob = 1 / (1 + exp(-linear_pred)) clamped to [1e-10, 1 - 1e-10];
    y#i ~ Bernoulli(prob) observed.  Every observe statement's parameter is an expression of the coefficients: no record stream, the
    model runs on the interpreter kernels (fg_hmc_interp.hip, fg_mh_interp.hip)."""
    from .model import Bernoulli, as_expr, exp
    n, p = X.shape
    P = Program()
    betas = [P.sample(addr("beta", j), Normal(0.0, 2.0)) for j in range(p)]
    for i in range(n):
        lp = as_expr(0.0)
        for j in range(p):
            lp = lp + betas[j] * float(X[i, j])
        prob = (1.0 / (1.0 + exp(-lp))).clamp(1e-10, 1.0 - 1e-10)
        P.observe(addr("y", i), Bernoulli(prob), bool(labels[i]))
    return P


def classification_data(n: int = 100, seed: int = 42):
    """The shape of classification.rs:36-60's generator: features [1, x1, x2], x ~ N(0, 1), true coefficients (-1.0, 2.0, -1.5)."""
    rng = np.random.default_rng(seed)
    X = np.column_stack([np.ones(n), rng.standard_normal(n), rng.standard_normal(n)])
    beta = np.array([-1.0, 2.0, -1.5])
    labels = rng.random(n) < 1.0 / (1.0 + np.exp(-(X @ beta)))
    return X, labels, beta


def ridge_data(n: int = 1024, p: int = 32, sigma: float = 0.5, seed_x: int = 7, seed_b: int = 8):
    X = np.random.default_rng(seed_x).standard_normal((n, p))
    beta = np.random.default_rng(seed_b).standard_normal(p)
    y = X @ beta + sigma * np.random.default_rng(seed_b + 1).standard_normal(n)
    return X, y, beta


def ridge_truth(X, y, sigma=0.5, lam=1.0):
    p = X.shape[1]
    Sigma = np.linalg.inv(X.T @ X / sigma ** 2 + lam * np.eye(p))
    return Sigma @ X.T @ y / sigma ** 2, Sigma


def smc_normal(y: float = 1.5) -> Program:
    """C4: /root/reference/examples/smc_inference.rs:36-39.  Posterior N(1.2, 0.2),
    log Z = log N(1.5; 0, sqrt(1.25)) = -1.9305103088617774."""
    return readme_normal(y=y, sigma=0.5)


def mixture(data: np.ndarray, K: int = 4) -> Program:
    """C5: mu#k ~ N(0,5); z#i ~ Categorical([1/K]*K); x#i ~ N(mu#z_i, 1) (pattern of
    /root/reference/examples/mixture_models.rs:77-112 with K components and fixed weights)."""
    P = Program()
    mus = [P.sample(addr("mu", k), Normal(0.0, 5.0)) for k in range(K)]
    for i, xi in enumerate(np.asarray(data, dtype=float)):
        z = P.sample(addr("z", i), Categorical([1.0 / K] * K))
        P.observe(addr("x", i), Normal(select(z, mus), 1.0), float(xi))
    return P


def mixture_data(n: int = 64, means=(-6.0, -2.0, 2.0, 6.0), seed: int = 9):
    rng = np.random.default_rng(seed)
    comp = rng.integers(0, len(means), size=n)
    return np.asarray(means)[comp] + rng.standard_normal(n), comp


def gamma_scale_model() -> Program:
    """Non-conjugate positive-support example used by the proposal tests
    (/root/reference/tests/f_mcmc_proposals.rs:31-70): x ~ Gamma(3,2), mean 1.5."""
    return trace_model(lambda: sample(addr("x"), Gamma(3.0, 2.0)))
