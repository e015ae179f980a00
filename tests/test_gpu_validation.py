"""The validation harness (fugue_amd.validation, mirror of src/inference/validation.rs) driven by the engine:
the reference's own acceptance procedure applied to many chains."""
import numpy as np
import pytest

from fugue_amd import engine as E
from fugue_amd import model as M
from fugue_amd import validation as V
from fugue_amd import workloads as W

pytestmark = pytest.mark.gpu


def _site_draws(cp, site, C, run):
    """run(engine, recorded sites, device buffer) -> draws [n][C] of `site`."""
    def fn(n_samples, n_warmup):
        eng = E.Engine(cp, C, seed=17)
        buf = eng.device_alloc(n_samples * max(1, cp.d) * C * 8)
        out = run(eng, n_samples, n_warmup, buf)
        eng.device_free(buf)
        return out
    return fn


def test_conjugate_normal_model_hmc_and_mh():
    """validation.rs:92-118 with the README model (prior N(0,1), likelihood sigma 0.5, y = 1.2): HMC and MH."""
    cp = E.compile_model(W.readme_normal())
    cfg = V.ConjugateNormalConfig(0.0, 1.0, 0.5, 1.2, n_samples=600, n_warmup=300)
    C = 512

    def hmc(eng, ns, nw, buf):
        eng.hmc_run(E.hmc_config(), ns, nw, buf)
        return eng.download(buf, (ns, 1, C))[:, 0, :]

    def mh(eng, ns, nw, buf):
        eng.mh_run(ns, nw, None, [0], buf)
        return eng.download(buf, (ns, 1, C))[:, 0, :]

    r = V.test_conjugate_normal_model(_site_draws(cp, "mu", C, hmc), cfg)
    # per chain each 2-sigma check passes ~95 % of the time for i.i.d. draws; HMC's super-efficient draws (ESS > n
    # for the mean) shrink the standard error the rule uses, so the per-chain pass rate is lower -- the pooled verdict,
    # over 307 200 draws, is the sharp one
    assert r.fraction_valid > 0.5 and r.pooled.is_valid(), r.pooled.summary()
    assert np.mean([x.ess_adequate for x in r.per_chain]) > 0.95
    r = V.test_conjugate_normal_model(_site_draws(cp, "mu", C, mh), cfg)
    assert r.pooled.mean_within_bounds and r.pooled.var_within_bounds, r.pooled.summary()
    assert np.mean([x.mean_within_bounds for x in r.per_chain]) > 0.85


def test_conjugate_beta_bernoulli_model_mh():
    """validation.rs:144-162: theta ~ Beta(2,2), 10 Bernoulli observations with 7 successes -> Beta(9,5)."""
    obs = [True, False, True, True, False, True, True, False, True, True]
    P = M.Program()
    th = P.sample(M.addr("theta"), M.Beta(2.0, 2.0))
    for i, o in enumerate(obs):
        P.observe(M.addr("flip", i), M.Bernoulli(th), 1.0 if o else 0.0)
    cp = E.compile_model(P)
    C = 512

    def mh(eng, ns, nw, buf):
        eng.mh_run(ns, nw, None, [0], buf)
        return eng.download(buf, (ns, 1, C))[:, 0, :]

    r = V.test_conjugate_beta_bernoulli_model(_site_draws(cp, "theta", C, mh), V.ConjugateBetaBernoulliConfig(2.0, 2.0, obs, 800, 400))
    assert r.pooled.mean_within_bounds and r.pooled.var_within_bounds, r.pooled.summary()
    assert np.mean([x.mean_within_bounds for x in r.per_chain]) > 0.85
    assert abs(r.pooled.posterior_mu - 9.0 / 14.0) < 1e-12


@pytest.mark.parametrize("name,dist,ref", [
    ("normal", lambda: M.Normal(1.0, 2.0), lambda g, n: 1.0 + 2.0 * g.standard_normal(n)),
    ("gamma", lambda: M.Gamma(3.0, 2.0), lambda g, n: g.gamma(3.0, 0.5, n)),
    ("beta", lambda: M.Beta(2.0, 5.0), lambda g, n: g.beta(2.0, 5.0, n)),
    ("lognormal", lambda: M.LogNormal(0.2, 0.7), lambda g, n: g.lognormal(0.2, 0.7, n)),
    ("exponential", lambda: M.Exponential(1.5), lambda g, n: g.exponential(1.0 / 1.5, n)),
    ("student_t", lambda: M.StudentT(5.0, 0.0, 1.0), lambda g, n: g.standard_t(5.0, n)),
])
def test_ks_test_distribution_on_prior_sampler(name, dist, ref):
    """validation.rs:17-44 (the pattern of tests/f_tests_sampler_validation.rs:348-648, alpha = 0.001): the engine's
    prior sampler against numpy's sampler of the same law, 20 000 draws each."""
    P = M.Program()
    P.sample(M.addr("x"), dist())
    eng = E.Engine(E.compile_model(P), 20000, seed=33)
    eng.prior_init()
    ours = eng.get_values().view(np.float64)[0]
    assert V.ks_test_distribution(ours, ref(np.random.default_rng(77), 20000), 0.001)
    assert not V.ks_test_distribution(ours * 1.1 + 0.05, ref(np.random.default_rng(78), 20000), 0.001)
