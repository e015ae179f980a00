"""The reference's own posterior-recovery tests as model builders + closed-form targets, shared by
the oracle (CPU) and engine (GPU) test files.  Each cites the reference test it restates."""
import math

import numpy as np

from fugue_amd import model as M


def conj_normal_5obs():
    """tests/f_hmc_posterior.rs:85-130: mu ~ N(0,2); y_i ~ N(mu,1), data [1,2,3,1.5,2.5].
    Posterior N(1.9047619, 0.19047619)."""
    data = [1.0, 2.0, 3.0, 1.5, 2.5]
    def model():
        return M.sample(M.addr("mu"), M.Normal(0.0, 2.0)).bind(
            lambda mu: M.plate(range(len(data)), lambda i: M.observe(M.addr("y", i), M.Normal(mu, 1.0), data[i])).map(lambda _: mu))
    return M.trace_model(model), 10.0 / 5.25, 1.0 / 5.25


def gamma31():
    """tests/f_hmc_posterior.rs:137-157: g ~ Gamma(3,1): support (0, inf), mean 3."""
    return M.trace_model(lambda: M.sample(M.addr("g"), M.Gamma(3.0, 1.0)))


def correlated_gaussian(rho=0.8):
    """tests/f_hmc_posterior.rs:32-80: x ~ N(0,1); y ~ N(rho x, sqrt(1-rho^2)): cov [[1,rho],[rho,1]]."""
    return M.trace_model(lambda: M.sample(M.addr("x"), M.Normal(0.0, 1.0)).bind(
        lambda x: M.sample(M.addr("y"), M.Normal(rho * x, math.sqrt(1 - rho * rho)))))


def axis_scaled():
    """src/inference/hmc.rs:996-1020: x ~ N(0,1), y ~ N(0,10) with adapt_mass."""
    return M.trace_model(lambda: M.sample(M.addr("x"), M.Normal(0.0, 1.0)).bind(lambda x: M.sample(M.addr("y"), M.Normal(0.0, 10.0))))


def beta_bernoulli():
    """tests/f_smc_smc.rs:25-65: theta ~ Beta(2,3)?  The reference states the posterior Beta(20,11) from 18 successes
    in 26 trials on a Beta(2,3) prior; restated with that data."""
    P = M.Program()
    th = P.sample(M.addr("theta"), M.Beta(2.0, 3.0))
    for i in range(26):
        P.observe(M.addr("y", i), M.Bernoulli(th), i < 18)
    return P, 20.0 / 31.0


def smc_5obs():
    """tests/f_smc_smc.rs:137-205: mu ~ N(0,1); y_j ~ N(mu,1), ys = [1,2,1.5,0.5,1.8]: log Z = -7.007239, mean 1.133333."""
    ys = [1.0, 2.0, 1.5, 0.5, 1.8]
    P = M.Program()
    mu = P.sample(M.addr("mu"), M.Normal(0.0, 1.0))
    for i, y in enumerate(ys):
        P.observe(M.addr("y", i), M.Normal(mu, 1.0), y)
    return P, -7.007239, 1.133333


def categorical_k(K=8):
    """tests/f_mcmc_proposals.rs:95-170: z ~ Categorical(uniform K); y ~ N(z, 1) observed at K-2:
    posterior over z proportional to N(K-2; z, 1)."""
    P = M.Program()
    z = P.sample(M.addr("z"), M.Categorical([1.0 / K] * K))
    P.observe(M.addr("y"), M.Normal(z, 1.0), float(K - 2))
    w = np.exp(-0.5 * (np.arange(K) - (K - 2.0)) ** 2)
    return P, w / w.sum()


def poisson1():
    """tests/f_mcmc_proposals.rs:230-276: k ~ Poisson(1): P(k=0) = e^-1."""
    return M.trace_model(lambda: M.sample(M.addr("k"), M.Poisson(1.0)))


def discrete_uniform_mode():
    """tests/f_hmc_discrete_uniform.rs:103-142: k ~ DiscreteUniform(0,10); y ~ N(k, 1) observed at 7: mode 7."""
    P = M.Program()
    k = P.sample(M.addr("k"), M.DiscreteUniform(0, 10))
    P.observe(M.addr("y"), M.Normal(k, 1.0), 7.0)
    return P
