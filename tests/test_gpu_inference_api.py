"""The drivers under the reference's names (fugue_amd.inference), used the way the reference's tests use them
(tests/f_hmc_posterior.rs, tests/f_mcmc_proposals.rs, tests/f_smc_smc.rs, tests/analytical_validation.rs)."""
import numpy as np
import pytest

import fugue_amd as F

pytestmark = pytest.mark.gpu


def conjugate_model():
    # prior N(0, 2), likelihood sigma 1, y = 3 -> posterior N(2.4, 0.8) (tests/analytical_validation.rs:24-70)
    return F.sample(F.addr("mu"), F.Normal(0.0, 2.0)).bind(lambda mu: F.observe(F.addr("y"), F.Normal(mu, 1.0), 3.0).map(lambda _: mu))


def test_hmc_chain_conjugate_normal():
    chains = F.hmc_chain(42, conjugate_model, 400, 300, F.HMCConfig(), n_chains=1024)
    mu = chains.get_f64(F.addr("mu"))
    assert mu.shape == (400, 1024)
    assert abs(mu.mean() - 2.4) < 0.01 and abs(mu.var() - 0.8) < 0.02
    assert 0.6 < chains.accept_rate <= 1.0 and chains.n_divergent == 0
    with pytest.raises(F.FugueError) as ei:
        chains.get_f64(F.addr("nope"))
    assert ei.value.code == 500                                       # TraceAddressNotFound


def test_adaptive_mcmc_chain_and_overrides():
    chains = F.adaptive_mcmc_chain(7, conjugate_model, 500, 500, n_chains=1024)
    mu = chains.get_f64(F.addr("mu"))
    assert abs(mu.mean() - 2.4) < 0.03 and 0.3 < chains.accept_rate < 0.6     # adapts towards 0.44
    # bounded site with a reflected walk (tests/f_mcmc_proposals.rs): a ~ U(0, 2) observed through N(a, 0.5) at 1.9
    model = lambda: F.sample(F.addr("a"), F.Uniform(0.0, 2.0)).bind(lambda a: F.observe(F.addr("y"), F.Normal(a, 0.5), 1.9).map(lambda _: a))
    ch = F.adaptive_mcmc_chain_with_overrides(3, model, 300, 300, [(F.addr("a"), F.SiteProposal.Reflect(0.0, 2.0))], n_chains=512)
    a = ch.get_f64(F.addr("a"))
    assert (a >= 0.0).all() and (a <= 2.0).all() and 1.2 < a.mean() < 1.8
    with pytest.raises(F.FugueError):
        F.adaptive_mcmc_chain_with_overrides(3, model, 10, 10, [(F.addr("zz"), F.SiteProposal.Gaussian())], n_chains=64)
    # discrete site through the same call
    coin = lambda: F.sample(F.addr("k"), F.Poisson(3.0)).map(lambda k: k)
    k = F.adaptive_mcmc_chain(5, coin, 400, 400, n_chains=1024).get_int(F.addr("k"))
    assert k.min() >= 0 and abs(k.mean() - 3.0) < 0.1


def test_adaptive_smc_evidence():
    # examples/smc_inference.rs model: log Z = log N(1.5; 0, sqrt(1.25))
    model = lambda: F.sample(F.addr("mu"), F.Normal(0.0, 1.0)).bind(lambda mu: F.observe(F.addr("y"), F.Normal(mu, 0.5), 1.5).map(lambda _: mu))
    r = F.adaptive_smc(42, 65536, model, F.SMCConfig(F.ResamplingMethod.Systematic, 0.5, 3))
    assert abs(r.log_evidence - (-1.9305103088617774)) < 0.02
    w = r.weights
    assert abs(w.sum() - 1.0) < 1e-9 and abs((w * r.get_f64(F.addr("mu"))).sum() - 1.2) < 0.02
    assert len(r.betas) >= 1 and r.betas[-1] == 1.0
    empty = F.adaptive_smc(1, 0, model, F.SMCConfig())
    assert empty.weights.size == 0 and empty.log_evidence == 0.0      # smc.rs:462-467


def test_hmc_chain_returns_every_site():
    """hmc_chain's states carry the whole trace: a discrete site keeps its prior draw next to the moving f64 site, and a
    model without continuous sites yields a fresh prior draw per step (hmc.rs:826-845)."""
    mixed = lambda: F.sample(F.addr("k"), F.Poisson(3.0)).bind(
        lambda k: F.sample(F.addr("mu"), F.Normal(0.0, 2.0)).bind(lambda mu: F.observe(F.addr("y"), F.Normal(mu, 1.0), 3.0).map(lambda _: mu)))
    ch = F.hmc_chain(11, mixed, 50, 50, F.HMCConfig(), n_chains=256)
    k, mu = ch.get_int(F.addr("k")), ch.get_f64(F.addr("mu"))
    assert k.shape == mu.shape == (50, 256)
    assert (k == k[0]).all() and k.min() >= 0 and 2.0 < k[0].mean() < 4.0
    assert abs(mu.mean() - 2.4) < 0.1
    coin = lambda: F.sample(F.addr("k"), F.Poisson(3.0)).map(lambda k: k)
    ch = F.hmc_chain(5, coin, 40, 10, F.HMCConfig(), n_chains=1024)
    k = ch.get_int(F.addr("k"))
    assert k.shape == (40, 1024) and (k[0] != k[1]).any() and abs(k.mean() - 3.0) < 0.1 and ch.accept_rate == 1.0
