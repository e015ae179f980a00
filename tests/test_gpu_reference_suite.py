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
