"""tests/f_tests_sampler_validation.rs (FG-14) against the engine's prior samplers: the reference's parameters, its one-sample
Kolmogorov-Smirnov test against the ANALYTIC cdf at alpha = 0.001 with the asymptotic critical value, its 5-standard-error moment
check, and its chi-square goodness of fit for the discrete distributions -- on 65 536 draws per distribution (the reference
draws 5 000: the bounds scale with n, so this is the stricter test)."""
import math

import numpy as np
import pytest
from scipy import stats

import fugue_amd as F
from fugue_amd import engine as E

pytestmark = pytest.mark.gpu
N, ALPHA = 65536, 0.001


def draw(dist, seed):
    eng = E.Engine(E.compile_model(lambda: F.sample(F.addr("x"), dist)), N, seed=seed)
    eng.prior_init()
    cells = eng.get_values()[0].copy()
    eng.close()
    return cells


def ks_critical(alpha, n):                                    # f_tests_sampler_validation.rs:287-289
    return math.sqrt(-0.5 * math.log(alpha / 2.0)) / math.sqrt(n)


def ks_one_sample(x, cdf):                                    # :271-282
    x = np.sort(x)
    n = len(x)
    f = cdf(x)
    i = np.arange(n)
    return max(((i + 1.0) / n - f).max(), (f - i / n).max())


CONTINUOUS = [   # name, distribution, scipy frozen distribution, (mean, sd) or None  -- parameters of :348-498
    ("Normal", lambda: F.Normal(2.0, 3.0), stats.norm(2.0, 3.0), (2.0, 3.0)),
    ("Uniform", lambda: F.Uniform(-3.0, 5.0), stats.uniform(-3.0, 8.0), (1.0, 8.0 / math.sqrt(12.0))),
    ("LogNormal", lambda: F.LogNormal(0.2, 0.5), stats.lognorm(0.5, scale=math.exp(0.2)), "scipy"),
    ("Exponential", lambda: F.Exponential(2.0), stats.expon(scale=0.5), (0.5, 0.5)),
    ("Beta", lambda: F.Beta(2.0, 5.0), stats.beta(2.0, 5.0), "scipy"),
    ("Gamma", lambda: F.Gamma(3.0, 2.0), stats.gamma(3.0, scale=0.5), (1.5, math.sqrt(3.0) / 2.0)),       # RATE parameterisation (:411-416)
    ("StudentT", lambda: F.StudentT(8.0, 1.0, 1.5), stats.t(8.0, 1.0, 1.5), "scipy"),
    ("Cauchy", lambda: F.Cauchy(0.5, 1.2), stats.cauchy(0.5, 1.2), None),                                   # KS only (:438)
    ("Laplace", lambda: F.Laplace(-1.0, 2.0), stats.laplace(-1.0, 2.0), "scipy"),
    ("Weibull", lambda: F.Weibull(1.5, 2.0), stats.weibull_min(1.5, scale=2.0), "scipy"),
    ("ChiSquared", lambda: F.ChiSquared(6.0), stats.chi2(6.0), (6.0, math.sqrt(12.0))),
    ("InverseGamma", lambda: F.InverseGamma(4.0, 3.0), stats.invgamma(4.0, scale=3.0), "scipy"),
]


@pytest.mark.parametrize("name,dist,ref,moments", CONTINUOUS, ids=[c[0] for c in CONTINUOUS])
def test_fg14_continuous_ks_and_moments(name, dist, ref, moments):
    x = draw(dist(), 1001).view(np.float64)
    assert np.isfinite(x).all()
    d = ks_one_sample(x, ref.cdf)
    assert d < ks_critical(ALPHA, N), f"{name}: D = {d:.5f} >= {ks_critical(ALPHA, N):.5f}"
    if moments is not None:
        mu, sd = (ref.mean(), ref.std()) if moments == "scipy" else moments
        z = abs(x.mean() - mu) / (sd / math.sqrt(N))
        assert z < 5.0, f"{name}: mean {x.mean():.5f} vs {mu:.5f}, z = {z:.2f}"


def chi_square(counts, probs):                                # :316-327
    e = np.asarray(probs) * counts.sum()
    return float(((counts - e) ** 2 / e).sum())


def moment_ok(mean, mu, sd):
    return abs(mean - mu) / (sd / math.sqrt(N)) < 5.0


def test_fg14_discrete_chi_square_and_moments():
    crit = lambda df: stats.chi2.ppf(1.0 - ALPHA, df)
    assert abs(crit(1) - 10.827566170662625) < 1e-9 and abs(crit(13) - 34.52817897487073) < 1e-9      # the reference's tabulated values (:332-343)
    # Bernoulli(0.3)  :504-520
    b = draw(F.Bernoulli(0.3), 2001)
    assert set(np.unique(b)) <= {0, 1}
    assert chi_square(np.bincount(b, minlength=2), [0.7, 0.3]) < crit(1) and moment_ok(b.mean(), 0.3, math.sqrt(0.21))
    # Categorical([0.1, 0.2, 0.3, 0.4])  :523-549
    probs = np.array([0.1, 0.2, 0.3, 0.4])
    c = draw(F.Categorical(list(probs)), 2002)
    mu = (np.arange(4) * probs).sum()
    sd = math.sqrt((np.arange(4) ** 2 * probs).sum() - mu * mu)
    assert c.min() >= 0 and c.max() <= 3
    assert chi_square(np.bincount(c, minlength=4), probs) < crit(3) and moment_ok(c.mean(), mu, sd)
    # Binomial(10, 0.4)  :552-582
    k = draw(F.Binomial(10, 0.4), 2003)
    assert k.min() >= 0 and k.max() <= 10
    assert chi_square(np.bincount(k, minlength=11), stats.binom.pmf(np.arange(11), 10, 0.4)) < crit(10) and moment_ok(k.mean(), 4.0, math.sqrt(2.4))
    # Poisson(4), bins 0..12 and the tail  :585-621
    p = draw(F.Poisson(4.0), 2004)
    pm = stats.poisson.pmf(np.arange(13), 4.0)
    assert p.min() >= 0
    assert chi_square(np.bincount(np.minimum(p, 13), minlength=14), np.append(pm, 1.0 - pm.sum())) < crit(13) and moment_ok(p.mean(), 4.0, 2.0)
    # DiscreteUniform(1, 6): a fair die  :624-650
    u = draw(F.DiscreteUniform(1, 6), 2005)
    assert u.min() >= 1 and u.max() <= 6
    assert chi_square(np.bincount(u - 1, minlength=6), [1.0 / 6] * 6) < crit(5) and moment_ok(u.mean(), 3.5, math.sqrt(35.0 / 12.0))
