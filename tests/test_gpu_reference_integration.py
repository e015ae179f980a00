"""The reference's own integration tests (tests/inference_integration.rs) written against the drivers and diagnostics under
the reference's names -- the same models, the same calls, the same assertions (tightened where many chains allow it).
One chain of the reference = one of the engine's chains; the pooled statements use all of them."""
import numpy as np
import pytest

import fugue_amd as F

pytestmark = pytest.mark.gpu
addr = F.addr


def test_mcmc_normal_mean_recovery():
    """inference_integration.rs:114-187: mu ~ N(0, 2), y ~ N(mu, 1) = 2.5 -> posterior N(2.0, 0.8)."""
    model = lambda: F.sample(addr("mu"), F.Normal(0.0, 2.0)).bind(lambda mu: F.observe(addr("y"), F.Normal(mu, 1.0), 2.5).bind(lambda _: F.pure(mu)))
    samples = F.adaptive_mcmc_chain(42, model, 1000, 200, n_chains=256)
    mu = samples.get_f64(addr("mu"))
    assert mu.shape == (1000, 256) and np.isfinite(mu).all()
    per_chain = mu.mean(axis=0)
    assert (np.abs(per_chain - 2.0) < 0.3).mean() > 0.95            # the reference's single-chain bound, for (almost) every chain
    assert abs(mu.mean() - 2.0) < 0.03 and abs(mu.var() - 0.8) < 0.05


def test_mcmc_beta_binomial_conjugacy():
    """inference_integration.rs:396-435: theta ~ Beta(2, 3), k ~ Binomial(10, clamp(theta)) = 7 -> Beta(9, 6)."""
    model = lambda: F.sample(addr("theta"), F.Beta(2.0, 3.0)).bind(
        lambda th: F.observe(addr("k"), F.Binomial(10, th.clamp(0.001, 0.999)), 7).bind(lambda _: F.pure(th)))
    samples = F.adaptive_mcmc_chain(42, model, 300, 50, n_chains=256)
    th = samples.get_f64(addr("theta"))
    assert th.shape == (300, 256) and np.isfinite(th).all() and (th >= 0.0).all() and (th <= 1.0).all()
    assert (np.abs(th.mean(axis=0) - 0.6) < 0.1).mean() > 0.9
    assert abs(th.mean() - 0.6) < 0.01 and th.var(axis=0).min() > 0.001


def test_smc_gaussian_model_and_resampling_methods():
    """inference_integration.rs:190-237, 438-481: 50 / 30 particles, no rejuvenation, all three resamplers."""
    model = lambda: F.sample(addr("mu"), F.Normal(0.0, 2.0)).bind(lambda mu: F.observe(addr("y"), F.Normal(mu, 0.5), 1.8).map(lambda _: mu))
    p = F.adaptive_smc(42, 50, model, F.SMCConfig(F.ResamplingMethod.Systematic, 0.5, 0))
    mu = p.get_f64(addr("mu"))
    assert mu.size == 50 and np.isfinite(p.log_weights).all()
    wm = (mu * np.exp(p.log_weights)).sum() / np.exp(p.log_weights).sum()
    assert -1.0 < wm < 3.0
    model2 = lambda: F.sample(addr("x"), F.Normal(0.0, 1.0)).bind(lambda x: F.observe(addr("y"), F.Normal(x, 0.5), 0.8).map(lambda _: x))
    for method in (F.ResamplingMethod.Systematic, F.ResamplingMethod.Multinomial, F.ResamplingMethod.Stratified):
        q = F.adaptive_smc(42, 30, model2, F.SMCConfig(method, 0.5, 0))
        assert q.weights.size == 30 and np.isfinite(q.log_weights).all()
        ess = F.effective_sample_size(q)
        assert 0.0 < ess <= 30.0 + 1e-9


def test_diagnostics_multi_chain():
    """inference_integration.rs:552-612: r_hat_f64 / summarize_f64_parameter / effective_sample_size_mcmc on a two-site model."""
    model = lambda: F.sample(addr("alpha"), F.Normal(0.0, 1.0)).bind(
        lambda a: F.sample(addr("beta"), F.Normal(a, 0.5)).map(lambda b: (a, b)))
    chains = F.adaptive_mcmc_chain(42, model, 100, 20, n_chains=3)
    for site in ("alpha", "beta"):
        r = F.r_hat_f64(chains, addr(site))
        assert np.isfinite(r) and r > 0.0
        s = F.summarize_f64_parameter(chains, addr(site))
        assert np.isfinite(s.mean) and np.isfinite(s.std) and s.std >= 0.0
        assert set(s.quantiles) == {"2.5%", "25%", "50%", "75%", "97.5%"} and s.quantiles["2.5%"] <= s.quantiles["50%"] <= s.quantiles["97.5%"]
        v = chains.get_f64(addr(site))[:, 0]
        ess = F.effective_sample_size_mcmc(v)
        assert 0.0 <= ess <= v.size


def test_workflow_complete_bayesian_analysis():
    """inference_integration.rs:671-756: prior -> MCMC (3 chains) -> diagnostics -> posterior predictive -> a model with no sites."""
    obs = [2.1, 1.8, 2.3, 1.9, 2.0]
    model = lambda: F.sample(addr("mu"), F.Normal(0.0, 2.0)).bind(
        lambda mu: F.sequence_vec([F.observe(addr("y", i), F.Normal(mu, 1.0), y) for i, y in enumerate(obs)]).map(lambda _: mu))
    chains = F.adaptive_mcmc_chain(42, model, 200, 50, n_chains=3)
    r = F.r_hat_f64(chains, addr("mu"))
    s = F.summarize_f64_parameter(chains, addr("mu"))
    assert np.isfinite(r) and r > 0.0 and np.isfinite(s.mean) and s.std > 0.0
    obs_mean = float(np.mean(obs))
    assert abs(s.mean - obs_mean) < 0.5
    post = chains.get_f64(addr("mu")).T.ravel()[:100]
    pred = post + np.random.default_rng(42).standard_normal(100)                 # Normal(mu, 1).sample per posterior draw
    assert np.isfinite(pred).all() and abs(pred.mean() - obs_mean) < 1.0
    simple = lambda: F.observe(addr("y", 0), F.Normal(2.0, 1.0), 2.1).map(lambda _: 2.0)     # no sample site at all
    simple_samples = F.adaptive_mcmc_chain(42, simple, 50, 10, n_chains=3)
    assert simple_samples.cells.shape[0] == 50


def test_workflow_parameter_estimation_uncertainty():
    """inference_integration.rs:763-860 (FG-49): linear regression with the exact bivariate-normal posterior; the bound is the
    reference's 4 standard errors from the chain's own ESS, per chain, and 4 pooled standard errors for the pooled mean."""
    x, y = [1.0, 2.0, 3.0, 4.0, 5.0], [2.1, 4.2, 5.8, 8.1, 9.9]
    model = lambda: F.sample(addr("alpha"), F.Normal(0.0, 2.0)).bind(lambda a: F.sample(addr("beta"), F.Normal(0.0, 2.0)).bind(
        lambda b: F.sequence_vec([F.observe(addr("obs", i), F.Normal(a + b * xi, 1.0), yi) for i, (xi, yi) in enumerate(zip(x, y))]).map(lambda _: (a, b))))
    samples = F.adaptive_mcmc_chain(42, model, 800, 150, n_chains=64)
    truth = {"alpha": (0.2463016330451495, 0.8491834774255523), "beta": (1.9204610951008645, 0.08069164265129683)}
    for site, (m, v) in truth.items():
        d = samples.get_f64(addr(site))
        ok = 0
        for c in range(8):                                                        # the reference's single-chain assertion on 8 of the chains
            ess = F.effective_sample_size_mcmc(d[:, c])
            ok += abs(d[:, c].mean() - m) < 4.0 * np.sqrt(v / ess)
        assert ok >= 7
        ess_all = F.effective_sample_size_multichain(d)
        assert abs(d.mean() - m) < 4.0 * np.sqrt(v / ess_all)
        assert abs(d.var() - v) < 0.35 * v


# ---- the reference's regression tests of proposals and SMC building blocks (tests/f_mcmc_proposals.rs, tests/f_smc_smc.rs) -------
def _mean_within_se(x, target, k):
    per_chain_ess = np.array([F.effective_sample_size_mcmc(x[:, c]) for c in range(min(8, x.shape[1]))])
    se = x[:, :len(per_chain_ess)].std(axis=0, ddof=1) / np.sqrt(per_chain_ess)
    return np.abs(x[:, :len(per_chain_ess)].mean(axis=0) - target) < k * se


def test_fg02_log_space_walk_targets_gamma_mean():
    """f_mcmc_proposals.rs:30-57, 61-82: x ~ Gamma(3, 2) sampled by the (auto-selected, then forced) log-space walk has mean 1.5
    (the Jacobian-less walk gave 1.0)."""
    auto = F.adaptive_mcmc_chain(20260710, lambda: F.sample(addr("x"), F.Gamma(3.0, 2.0)), 12000, 3000, n_chains=64).get_f64(addr("x"))
    assert _mean_within_se(auto, 1.5, 3.0).sum() >= 7 and (auto.mean(axis=0) > 1.25).all() and abs(auto.mean() - 1.5) < 0.01
    forced = F.adaptive_mcmc_chain_with_overrides(13371337, lambda: F.sample(addr("theta"), F.Gamma(3.0, 2.0)), 12000, 3000,
                                                  [(addr("theta"), F.SiteProposal.LogSpace())], n_chains=64).get_f64(addr("theta"))
    assert _mean_within_se(forced, 1.5, 3.0).sum() >= 7 and abs(forced.mean() - 1.5) < 0.01


def test_fg42_name_heuristic_no_longer_traps_unbounded_parameter():
    """f_mcmc_proposals.rs:89-113: an address literally named "p" with a Normal(0.5, 2) target is not confined to [0, 1]."""
    x = F.adaptive_mcmc_chain(31415, lambda: F.sample(addr("p"), F.Normal(0.5, 2.0)), 20000, 4000, n_chains=32).get_f64(addr("p"))
    outside = ((x < 0.0) | (x > 1.0)).mean(axis=0)
    assert (outside > 0.5).all() and (x.std(axis=0, ddof=1) > 1.0).all()
    assert abs(outside.mean() - 0.8026) < 0.01                                   # 1 - (Phi(0.25) - Phi(-0.25))


def test_fg10_categorical_top_categories_reachable_for_large_k():
    """f_mcmc_proposals.rs:224-251: K = 12 uniform prior, means 0..11, y = 10, sigma = 1.5: the whole posterior, top category included."""
    K = 12
    model = lambda: F.sample(addr("z"), F.Categorical([1.0 / K] * K)).bind(lambda z: F.observe(addr("y"), F.Normal(z, 1.5), 10.0).map(lambda _: z))
    z = F.adaptive_mcmc_chain(20260711, model, 8000, 800, n_chains=256).get_int(addr("z"))
    emp = np.bincount(z.ravel(), minlength=K) / z.size
    w = np.exp(-0.5 * ((10.0 - np.arange(K)) / 1.5) ** 2)
    expected = w / w.sum()
    assert abs(emp[11] - expected[11]) < 0.015 and np.abs(emp - expected).sum() < 0.03
    assert abs(expected[11] - 0.251748) < 1e-6                                   # the value the reference's test states


def _engine(model, n, seed):
    from fugue_amd import engine as E
    return E, E.Engine(E.compile_model(model), n, seed=seed)


def test_fg03_smc_prior_weights_do_not_square_the_prior():
    """f_smc_smc.rs:45-78: self-normalised importance weights of smc_prior_particles are the LIKELIHOOD (posterior mean of theta
    = 20/31), not prior x joint (27/45)."""
    def model():
        return F.sample(addr("theta"), F.Beta(2.0, 3.0)).bind(
            lambda th: F.sequence_vec([F.observe(addr("y", i), F.Bernoulli(th), i < 18) for i in range(26)]).map(lambda _: th))
    E, eng = _engine(model, 2000, 20260710)
    eng.smc_prior_particles()
    lw, w = eng.smc_weights()
    theta = eng.get_values()[0].view(np.float64)
    wm = float((w * theta).sum())
    assert abs(w.sum() - 1.0) < 1e-12
    assert abs(wm - 20.0 / 31.0) < 0.03 and abs(wm - 20.0 / 31.0) < abs(wm - 27.0 / 45.0)
    eng.close()


def test_fg13_rejuvenation_preserves_uniform_weights():
    """f_smc_smc.rs:83-140: resampling makes the weights uniform (ESS = N); an invariant MH rejuvenation sweep moves particles and
    leaves the weights alone."""
    model = lambda: F.sample(addr("mu"), F.Normal(0.0, 1.0)).bind(lambda mu: F.observe(addr("y"), F.Normal(mu, 1.0), 1.0).map(lambda _: mu))
    E, eng = _engine(model, 50, 7)
    eng.smc_prior_particles()
    eng.smc_resample(E.RESAMPLE_SYSTEMATIC)
    assert abs(eng.smc_ess() - 50.0) < 1e-9
    before = eng.get_values()[0].view(np.float64).copy()
    eng.smc_rejuvenate(1.0, 5)
    assert abs(eng.smc_ess() - 50.0) < 1e-9
    _, w = eng.smc_weights()
    assert np.abs(w - 1.0 / 50).max() < 1e-12
    after = eng.get_values()[0].view(np.float64)
    assert (np.abs(before - after) > 1e-9).any()
    eng.close()
