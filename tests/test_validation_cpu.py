"""fugue_amd.validation (mirror of src/inference/validation.rs) on the CPU: the KS statistic and threshold, the
scoring rule's arithmetic, the conjugate posteriors, and the single-chain ESS against the oracle's restatement."""
import numpy as np

from fugue_amd import validation as V


def test_ks_statistic_and_threshold():
    rng = np.random.default_rng(50)
    a, b = rng.standard_normal(200), rng.standard_normal(200)
    assert V.ks_test_distribution(a, b, 0.05)                      # validation.rs:325-331: same law passes
    assert not V.ks_test_distribution(a + 1.0, b, 0.05)            # a shifted sample does not
    # literal walk: identical samples differ by at most one step of the slower ECDF
    s = np.sort(a)
    assert V.ks_statistic(s, s) <= 1.0 / 200 + 1e-15
    assert abs(V.ks_statistic(np.array([0.0, 1.0]), np.array([2.0, 3.0])) - 0.5) < 1e-15   # walk stops when sample1 is exhausted
    crit = np.sqrt(-0.5 * np.log(0.001)) * np.sqrt(400 / 40000)
    assert abs(crit - 0.18585) < 1e-4


def test_conjugate_posteriors():
    mu, var = V.ConjugateNormalConfig(0.0, 1.0, 0.5, 1.2, 1000, 500).posterior()
    assert abs(mu - 0.96) < 1e-12 and abs(var - 0.2) < 1e-12     # README model: N(0.96, 0.2)
    mu, var = V.ConjugateBetaBernoulliConfig(2.0, 2.0, [True] * 7 + [False] * 3, 100, 10).posterior()
    assert abs(mu - 9.0 / 14.0) < 1e-12 and abs(var - 9.0 * 5.0 / (14.0 * 14.0 * 15.0)) < 1e-12


def test_scoring_rule_on_iid_and_biased_draws(oracle):
    rng = np.random.default_rng(3)
    x = 0.96 + np.sqrt(0.2) * rng.standard_normal(2000)
    r = V.validate_against_analytical_posterior(x, 0.96, 0.2, 2000)
    assert abs(r.effective_sample_size - oracle.ess_single(x)) < 1e-6 * r.effective_sample_size   # == effective_sample_size_mcmc
    assert r.ess_adequate and r.mean_within_bounds and r.var_within_bounds and r.is_valid()
    assert "Overall: PASS" in r.summary() and "True posterior: N(0.9600, 0.4472)" in r.summary()
    bad = V.validate_against_analytical_posterior(x + 0.2, 0.96, 0.2, 2000)
    assert not bad.mean_within_bounds and not bad.is_valid() and "Mean error" in bad.summary()
    assert V.validate_against_analytical_posterior(np.array([]), 0.0, 1.0, 10).failed == "No samples extracted"
    assert V.effective_sample_size_mcmc(np.array([1.0, 2.0, 3.0])) == 3.0                         # n < 4
    # sticky chain: 10 % efficiency rule fails
    sticky = np.repeat(rng.standard_normal(40), 50)
    assert not V.validate_against_analytical_posterior(sticky, 0.0, 1.0, 2000).ess_adequate


def test_batch_validation_over_chains():
    rng = np.random.default_rng(5)
    draws = 0.96 + np.sqrt(0.2) * rng.standard_normal((400, 64))
    b = V.validate_chains(draws, 0.96, 0.2, 400)
    assert len(b.per_chain) == 64 and b.fraction_valid > 0.8 and b.pooled.is_valid()
    assert b.pooled.effective_sample_size > 0.5 * 400 * 64
