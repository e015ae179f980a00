"""Pins the oracle's control flow with the reference's own deterministic unit tests and
posterior-recovery tests, restated (same models, same tolerances) -- see tests/reference_suite.py."""
import math

import numpy as np
import pytest

from fugue_amd import model as M
from fugue_amd import workloads as W
from tests import reference_suite as R


def _f64(cells):
    return np.ascontiguousarray(cells).view(np.float64)


def test_dual_averaging_moves_step_size_toward_target(oracle):
    """src/inference/hmc.rs:934-957"""
    _, _, frozen = oracle.dual_averaging(1.0, 0.8, [0.1] * 200)
    assert frozen < 1.0
    _, _, frozen = oracle.dual_averaging(1.0, 0.8, [1.0] * 200)
    assert frozen > 1.0


def test_log_joint_is_sum_of_log_pdfs(oracle):
    """src/inference/hmc.rs:975-989: mu ~ N(0,1); y ~ N(mu,1) = 1.0"""
    om = oracle.OracleModel(M.trace_model(lambda: M.sample(M.addr("mu"), M.Normal(0.0, 1.0)).bind(
        lambda mu: M.observe(M.addr("y"), M.Normal(mu, 1.0), 1.0))))
    for mu in (-1.3, 0.0, 0.7):
        acc, logp = om.run_score(om.cells([mu]))
        exp = oracle.logpdf("Normal", mu, [0.0, 1.0]) + oracle.logpdf("Normal", 1.0, [mu, 1.0])
        assert abs(acc.sum() - exp) < 1e-9 and abs(logp[0] - acc[0]) < 1e-15     # fresh logp == log_prior (f_runtime_audit.rs:411-446)


def test_hmc_standard_normal_marginal(oracle):
    """src/inference/hmc.rs:960-970"""
    om = oracle.OracleModel(M.trace_model(lambda: M.sample(M.addr("x"), M.Normal(0.0, 1.0))))
    draws, _, _, st = om.hmc_run(7, 1, 500, 2000)
    x = draws[:, 0, 0]
    assert abs(x.mean()) < 0.1 and abs(x.var() - 1.0) < 0.15


def test_hmc_conjugate_normal_matches_analytic(oracle):
    """tests/f_hmc_posterior.rs:85-130 (one chain, 1000 + 4000, 3 SE / 12 %)"""
    prog, pm, pv = R.conj_normal_5obs()
    om = oracle.OracleModel(prog)
    draws, _, _, _ = om.hmc_run(4242, 1, 1000, 4000)
    mu = draws[:, 0, 0]
    se = math.sqrt(mu.var() / oracle.ess_single(mu))
    assert abs(mu.mean() - pm) < 3 * se and abs(mu.var() - pv) < 0.12 * pv


def test_hmc_bounded_support_stays_in_support(oracle):
    """tests/f_hmc_posterior.rs:137-157"""
    om = oracle.OracleModel(R.gamma31())
    draws, _, _, _ = om.hmc_run(99, 1, 1000, 3000)
    g = draws[:, 0, 0]
    assert (g > 0).all() and np.isfinite(g).all() and abs(g.mean() - 3.0) < 0.3


def test_hmc_mass_adaptation_axis_scaled(oracle):
    """src/inference/hmc.rs:996-1020"""
    om = oracle.OracleModel(R.axis_scaled())
    draws, _, _, _ = om.hmc_run(5, 4, 600, 1500, oracle.HmcConfig.default(adapt_mass=1))
    assert abs(draws[:, 0].std() - 1.0) < 0.2 and abs(draws[:, 1].std() - 10.0) < 2.0


def test_mh_one_model_run_per_step_and_frozen_scales(oracle):
    """src/inference/mh.rs:1186-1202 (one run per step), :1207-1290 (scales frozen after warmup)"""
    om = oracle.OracleModel(W.reference_model(6))
    _, _, s1, st1 = om.mh_run(3, 4, 200, 0)
    _, _, s2, st2 = om.mh_run(3, 4, 200, 300)
    assert st1.n_model_evals == 4 * 200 and st2.n_model_evals == 4 * 500
    assert np.array_equal(s1, s2)


def test_mh_log_space_jacobian_identity(oracle):
    """src/inference/mh.rs:1067-1079: q(x'|x) - q(x|x') + ... the correction reduces to ln x' - ln x.
    Checked through the sampler: a Gamma(3,2) chain with the log-space walk has mean 1.5
    (tests/f_mcmc_proposals.rs:31-70)."""
    om = oracle.OracleModel(W.gamma_scale_model())
    draws, _, _, _ = om.mh_run(11, 64, 500, 1500)
    x = _f64(draws[:, 0, :])
    assert (x > 0).all() and abs(x.mean() - 1.5) < 0.05


def test_mh_categorical_and_count_posteriors(oracle):
    """tests/f_mcmc_proposals.rs:95-276: K = 8 categorical posterior L1 < 0.03; Poisson(1): P(0) = e^-1 +- 0.03"""
    prog, post = R.categorical_k(8)
    om = oracle.OracleModel(prog)
    draws, _, _, _ = om.mh_run(5, 64, 200, 800)
    freq = np.bincount(draws[:, 0, :].ravel(), minlength=8) / draws[:, 0, :].size
    assert np.abs(freq - post).sum() < 0.03
    om = oracle.OracleModel(R.poisson1())
    draws, _, _, _ = om.mh_run(6, 64, 300, 800)
    assert abs((draws[:, 0, :] == 0).mean() - math.exp(-1)) < 0.03
    om = oracle.OracleModel(R.discrete_uniform_mode())
    draws, _, _, _ = om.mh_run(7, 64, 300, 500)
    assert np.bincount(draws[:, 0, :].ravel()).argmax() == 7        # tests/f_hmc_discrete_uniform.rs:103-142


def test_mh_coin_flip_config1(oracle):
    """BASELINE configs[0]: bayesian_coin_flip.rs, adaptive_mcmc_chain 500 + 1000, 1 chain: Beta(9,5), mean 0.642857"""
    om = oracle.OracleModel(W.coin_flip())
    draws, _, _, _ = om.mh_run(42, 1, 500, 1000)
    p = _f64(draws[:, 0, 0])
    assert abs(p.mean() - 9 / 14) < 0.03 and (0 < p).all() and (p < 1).all()


def test_resampler_shapes_and_uniform_weights_after_rejuvenation(oracle):
    """src/inference/smc.rs:801-847 (index range / count), tests/f_smc_smc.rs:66-135 (ESS == N after an
    invariant move: weights are not touched by rejuvenation)"""
    w = np.array([0.1, 0.2, 0.3, 0.4])
    for idx in (oracle.systematic_indices(w, 0.3), oracle.stratified_indices(w, [0.1, 0.5, 0.9, 0.2]),
                oracle.multinomial_indices(w, [0.05, 0.35, 0.61, 0.99])):
        assert len(idx) == 4 and idx.min() >= 0 and idx.max() < 4
    assert oracle.ess_particles(np.full(4, 0.25)) == pytest.approx(4.0)
    assert oracle.ess_particles([0.99, 0.01]) < 1.1                       # smc.rs:222-229 doc example


def test_smc_prior_weights_do_not_square_the_prior(oracle):
    """tests/f_smc_smc.rs:45-65: importance weights = likelihood only; mean of Beta(20,11) within 0.03"""
    prog, mean = R.beta_bernoulli()
    r = oracle.OracleModel(prog).smc_run(2000, 20260710, rejuvenation_steps=0)
    est = float((r["weights"] * r["values"].view(np.float64)[0]).sum())
    assert abs(est - mean) < 0.03 and abs(est - mean) < abs(est - 27 / 45)


@pytest.mark.parametrize("batched", [0, 1])
def test_tempered_smc_matches_conjugate_evidence_and_mean(oracle, batched):
    """tests/f_smc_smc.rs:137-205: N = 2000, rejuvenation 3: log Z = -7.007239 +- 0.2, mean 1.133333 +- 0.06;
    the per-sweep batched adaptation (GPU semantics) meets the same bars as the sequential one."""
    prog, logz, mean = R.smc_5obs()
    r = oracle.OracleModel(prog).smc_run(2000, 2026, rejuvenation_steps=3, batched=batched)
    est = float((r["weights"] * r["values"].view(np.float64)[0]).sum() / r["weights"].sum())
    assert abs(est - mean) < 0.06 and abs(r["log_evidence"] - logz) < 0.2
    assert r["betas"][-1] == 1.0 and np.all(np.diff(r["betas"]) > 0)


def test_smc_example_config4(oracle):
    """examples/smc_inference.rs:44-110: N(1.2, 0.2), |mean err| < 0.15, |var err| < 0.1"""
    r = oracle.OracleModel(W.smc_normal()).smc_run(2000, 42, rejuvenation_steps=3)
    mu = r["values"].view(np.float64)[0]
    m = float((r["weights"] * mu).sum())
    v = float((r["weights"] * (mu - m) ** 2).sum())
    assert abs(m - 1.2) < 0.15 and abs(v - 0.2) < 0.1 and math.isfinite(r["log_evidence"])


def test_guard_strictness_on_values_with_positive_probability_of_equality(oracle):
    """model.rs:710-716: guard(bool) accepts exactly when the predicate holds.  `k >= 1` on a Poisson site must accept k == 1
    and `k > 1` must reject it; a clamped expression sitting ON its bound satisfies `>=` and fails `>`."""
    def build(pred):
        return M.trace_model(lambda: M.sample(M.addr("k"), M.Poisson(2.0)).bind(
            lambda k: M.sample(M.addr("s"), M.Normal(0.0, 1.0)).bind(lambda s: M.guard(pred(k, s)).map(lambda _: k))))
    cases = {
        "k>=1": (lambda k, s: k >= 1, {0: False, 1: True, 2: True}),
        "k>1": (lambda k, s: k > 1, {0: False, 1: False, 2: True}),
        "k<=1": (lambda k, s: k <= 1, {0: True, 1: True, 2: False}),
        "k<1": (lambda k, s: k < 1, {0: True, 1: False, 2: False}),
        "clamped>=": (lambda k, s: s.max(0.01) >= 0.01, {0: True, 1: True, 2: True}),       # s = -1 below: the clamped value sits on the bound
        "clamped>": (lambda k, s: s.max(0.01) > 0.01, {0: False, 1: False, 2: False}),
    }
    for name, (pred, expect) in cases.items():
        om = oracle.OracleModel(build(pred))
        assert om.site_names == ["k", "s"]
        for kv, ok in expect.items():
            cells = np.array([kv, np.float64(-1.0).view(np.int64)], dtype=np.int64)
            acc, _ = om.run_score(cells)
            assert (acc[2] == 0.0) if ok else (acc[2] == -math.inf), (name, kv, acc)
    # constant predicates (a literal margin): equality passes `>=` and fails `>`
    assert len(M.trace_model(lambda: M.guard(M.Cond(M.as_expr(0.0), strict=False))).stmts) == 0
    assert len(M.trace_model(lambda: M.guard(M.Cond(M.as_expr(0.0), strict=True))).stmts) == 1
