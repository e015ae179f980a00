"""The reference's posterior-recovery tests (tests/reference_suite.py) run on the GPU engine with
thousands of chains: same models, the reference's tolerances (met with a wide margin)."""
import math

import numpy as np
import pytest

from fugue_amd import diagnostics as D
from fugue_amd import engine as E
from fugue_amd import workloads as W
from tests import reference_suite as R

pytestmark = pytest.mark.gpu


def _hmc(prog, C, nw, ns, seed=1, **cfg):
    cp = E.compile_model(prog)
    eng = E.Engine(cp, C, seed=seed)
    d = eng.device_alloc(ns * cp.d * C * 8)
    st = eng.hmc_run(E.hmc_config(**cfg), ns, nw, d)
    draws = eng.download(d, (ns, cp.d, C))
    return cp, eng, st, draws, d


def _mh(prog, C, nw, ns, seed=1, overrides=None):
    cp = E.compile_model(prog)
    eng = E.Engine(cp, C, seed=seed)
    d = eng.device_alloc(ns * cp.S * C * 8)
    st = eng.mh_run(ns, nw, overrides, list(range(cp.S)), d)
    return cp, st, eng.download(d, (ns, cp.S, C), dtype=np.int64)


@pytest.mark.parametrize("mode", [E.GRAD_FD_DENSE, E.GRAD_FD_SPARSE])
def test_hmc_conjugate_normal(mode):
    prog, pm, pv = R.conj_normal_5obs()
    cp, eng, st, draws, d = _hmc(prog, 2048, 300, 300, grad_mode=mode)
    assert abs(draws.mean() - pm) < 5e-3 and abs(draws.var() - pv) < 0.03 * pv
    cd = D.ChainDiagnostics(D.EngineMoments(eng, d, 300, cp.d))
    assert cd.split_rhat()[0] < 1.01 and st.n_divergent == 0


def test_hmc_bounded_support_and_correlated_target():
    cp, eng, st, draws, _ = _hmc(R.gamma31(), 2048, 300, 300)
    assert (draws > 0).all() and abs(draws.mean() - 3.0) < 0.05        # non-hoistable? Gamma(3,1) constant params: hoisted lgamma
    cp, eng, st, draws, _ = _hmc(R.correlated_gaussian(0.8), 2048, 300, 300)
    x, y = draws[:, 0].ravel(), draws[:, 1].ravel()
    cov = np.cov(np.stack([x, y]))
    assert np.abs(cov - np.array([[1.0, 0.8], [0.8, 1.0]])).max() < 0.15 * 1.0     # tests/f_hmc_posterior.rs:32-80 (15 %)


def test_hmc_mass_adaptation_axis_scaled():
    cp, eng, st, draws, _ = _hmc(R.axis_scaled(), 1024, 400, 400, adapt_mass=True)
    assert abs(draws[:, 0].std() - 1.0) < 0.1 and abs(draws[:, 1].std() - 10.0) < 1.0
    m_inv = eng.hmc_mass()
    assert np.median(m_inv[1]) > 20 * np.median(m_inv[0])               # the adapted mass sees the 100x variance ratio


def test_mh_reference_posteriors():
    prog, post = R.categorical_k(8)
    cp, st, draws = _mh(prog, 2048, 200, 400)
    freq = np.bincount(draws[:, 0, :].ravel(), minlength=8) / draws[:, 0, :].size
    assert np.abs(freq - post).sum() < 0.01
    cp, st, draws = _mh(R.poisson1(), 2048, 300, 400)
    assert abs((draws[:, 0, :] == 0).mean() - math.exp(-1)) < 0.01
    cp, st, draws = _mh(R.discrete_uniform_mode(), 2048, 300, 300)
    assert np.bincount(draws[:, 0, :].ravel()).argmax() == 7
    cp, st, draws = _mh(W.coin_flip(), 2048, 500, 500)                  # BASELINE configs[0] model, many chains
    p = draws[:, 0, :].view(np.float64)
    assert abs(p.mean() - 9 / 14) < 5e-3 and abs(p.var() - 0.0153061) < 2e-3


def test_smc_reference_targets():
    prog, logz, mean = R.smc_5obs()
    cp = E.compile_model(prog)
    r = E.Engine(cp, 65536, seed=2026).smc_run(rejuvenation_steps=3)
    est = float((r["weights"] * r["values"].view(np.float64)[0]).sum())
    assert abs(est - mean) < 0.01 and abs(r["log_evidence"] - logz) < 0.02
    prog, bmean = R.beta_bernoulli()
    r = E.Engine(E.compile_model(prog), 65536, seed=3).smc_run(rejuvenation_steps=0)
    assert abs(float((r["weights"] * r["values"].view(np.float64)[0]).sum()) - bmean) < 5e-3


def test_fg31_hmc_beats_mh_on_ess_per_model_eval():
    """tests/f_hmc_efficiency.rs:25-80: on a rho = 0.99 Gaussian, hmc_chain (L = 12, reference arithmetic: dense FD) must be at
    least 2x more efficient than adaptive_mcmc_chain PER MODEL EVALUATION for s = x + y.  Model evaluations are counted the
    way the reference counts model_fn calls: (L + 1) * 2d + 1 per HMC transition (hmc.rs:304-329, 353-407, 283-299; the
    Alg. 4 step-size search adds at most 100 * (2 * 2d + 1) once), one per MH step (mh.rs:1186-1202).  ESS is the mean
    over chains of effective_sample_size_mcmc."""
    from fugue_amd import validation as V
    prog = R.correlated_gaussian(0.99)
    C, L, d = 64, 12, 2
    cp, eng, st, draws, _ = _hmc(prog, C, 600, 1500, seed=20260711, n_leapfrog=L, grad_mode=E.GRAD_FD_DENSE)
    s = draws[:, 0, :] + draws[:, 1, :]
    hmc_ess = np.mean([V.effective_sample_size_mcmc(s[:, c]) for c in range(C)])
    hmc_evals = (1500 + 600) * ((L + 1) * 2 * d + 1) + 100 * (2 * 2 * d + 1)
    cp, st2, cells = _mh(prog, C, 2000, 12000, seed=20260711)
    xy = cells.view(np.float64)
    s2 = xy[:, 0, :] + xy[:, 1, :]
    mh_ess = np.mean([V.effective_sample_size_mcmc(s2[:, c]) for c in range(C)])
    mh_evals = 12000 + 2000
    ratio = (hmc_ess / hmc_evals) / (mh_ess / mh_evals)
    assert ratio >= 2.0, (hmc_ess, hmc_evals, mh_ess, mh_evals, ratio)
