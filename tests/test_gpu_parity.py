"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs.  Integer / index results must match exactly; floating-point results
within the tolerances written beside each assertion."""
import numpy as np
import pytest

from fugue_amd import engine as E
from fugue_amd import workloads as W
from tests.models import ZOO, f64_values_for

pytestmark = pytest.mark.gpu


def _pair(oracle, name):
    prog = ZOO[name]()
    return E.compile_model(prog), oracle.OracleModel(prog)


def _f64(cells):
    return np.ascontiguousarray(cells).view(np.float64)


def _close(got, exp, rtol, atol=0.0):
    got, exp = np.asarray(got), np.asarray(exp)
    both_inf = np.isinf(got) & np.isinf(exp) & (np.sign(got) == np.sign(exp))
    both_nan = np.isnan(got) & np.isnan(exp)
    ok = both_inf | both_nan | (np.abs(got - exp) <= atol + rtol * np.abs(exp))
    assert ok.all(), (got[~ok][:5], exp[~ok][:5])


@pytest.mark.parametrize("name", list(ZOO))
def test_site_tables_agree(oracle, name):
    cp, om = _pair(oracle, name)
    assert cp.site_names == om.site_names            # BTreeMap (lexicographic) order
    assert cp.site_vtypes == om.site_vtypes
    assert cp.f64_sites == om.f64_sites


@pytest.mark.parametrize("name", list(ZOO))
def test_log_joint_matches_oracle(oracle, name):
    """ScoreGivenTrace on the GPU == oracle: per-site logp and the three accumulators.
    Tolerance 1e-12 relative: ocml vs glibc log/lgamma/pow differ by a few ulp."""
    cp, om = _pair(oracle, name)
    C = 200                                             # not a multiple of 64: exercises the tail wave
    cells = f64_values_for(om, np.random.default_rng(5), C)
    eng = E.Engine(cp, C, seed=1)
    eng.set_values(cells)
    acc, logp = eng.log_joint(want_logp=True)
    for c in range(C):
        oacc, ologp = om.run_score(cells[:, c])
        _close(logp[:, c], ologp, 1e-12, 1e-13)
        _close(acc[:, c], oacc, 1e-12, 1e-12)
    assert np.array_equal(eng.get_values(), cells)


@pytest.mark.parametrize("name", list(ZOO))
def test_prior_init_matches_oracle(oracle, name):
    """PriorHandler on the GPU == oracle draw for draw (shared Philox streams).  Discrete
    sites exact; f64 sites to 1e-12 (transcendentals in the samplers)."""
    cp, om = _pair(oracle, name)
    C = 130
    eng = E.Engine(cp, C, seed=77, chain_offset=1000)
    acc = eng.prior_init(iteration=3)
    got = eng.get_values()
    for c in range(C):
        cells, oacc, _ = om.run_prior(77, 1000 + c, it=3)
        for j in range(cp.S):
            if cp.site_vtypes[j] == 0:
                _close(_f64(got[j:j + 1, c]), _f64(cells[j:j + 1]), 1e-11, 1e-300)
            else:
                assert got[j, c] == cells[j], (name, c, cp.site_names[j])
        _close(acc[:, c], oacc, 1e-10, 1e-10)


@pytest.mark.parametrize("name", ["readme", "normal32", "coin", "refmodel8", "ridge", "mixture", "alldists"])
def test_fd_gradient_matches_oracle(oracle, name):
    """grad_log_joint (hmc.rs:304-329): dense FD vs oracle, and sparse FD vs dense.
    A central difference with h=1e-5 amplifies a 1-ulp difference in log pi by 1/(2h), so
    the tolerance is absolute 5e-6 * (1 + |lj|) -- the size of the reference's own FD noise."""
    cp, om = _pair(oracle, name)
    C = 70
    cells = f64_values_for(om, np.random.default_rng(11), C)
    eng = E.Engine(cp, C, seed=1)
    eng.set_values(cells)
    g_dense, ok_d = eng.hmc_grad(1e-5, E.GRAD_FD_DENSE)
    g_sparse, ok_s = eng.hmc_grad(1e-5, E.GRAD_FD_SPARSE)
    for c in range(C):
        q = _f64(cells[om.f64_sites, c])
        og, ook = om.grad_log_joint(cells[:, c], q)
        lj = abs(om.log_joint_at(cells[:, c], q))
        tol = 5e-6 * (1.0 + (lj if np.isfinite(lj) else 0.0))
        fin = np.isfinite(og)
        assert ook == bool(ok_d[c]) == bool(ok_s[c])
        assert np.array_equal(np.isfinite(g_dense[:, c]), fin)
        _close(g_dense[fin, c], og[fin], 1e-7, tol)
        _close(g_sparse[fin, c], og[fin], 1e-7, tol)


@pytest.mark.parametrize("name", ["readme", "normal32", "refmodel8", "ridge", "alldists"])
@pytest.mark.parametrize("mode", [E.GRAD_FD_DENSE, E.GRAD_FD_SPARSE])
def test_hmc_transition_injected(oracle, name, mode):
    """hmc_transition (hmc.rs:419-473) under injected momentum and uniform: accept decision,
    acceptance probability, divergence flag and the next state vs the oracle."""
    cp, om = _pair(oracle, name)
    C, rng = 96, np.random.default_rng(21)
    cells = f64_values_for(om, rng, C)
    p0 = rng.standard_normal((cp.d, C))
    u = rng.random(C)
    cfg = E.hmc_config(n_leapfrog=5, grad_mode=mode)
    eps = 0.05
    eng = E.Engine(cp, C, seed=1)
    eng.set_values(cells)
    acc, alpha, div, lj = eng.hmc_transition_injected(cfg, eps, p0, u)
    nxt = eng.get_values()
    n_flip = 0
    for c in range(C):
        q = _f64(cells[om.f64_sites, c])
        lj0 = om.log_joint_at(cells[:, c], q)
        qo, ljo, oacc, oalpha, odiv = om.hmc_transition(cells[:, c], q, lj0, eps, 5, p0[:, c], u[c])
        assert odiv == bool(div[c])
        if odiv:
            continue
        _close(alpha[c], oalpha, 1e-6, 1e-9)
        if oacc != bool(acc[c]):                       # only possible on a knife edge |u - alpha| ~ 1e-7
            assert abs(u[c] - oalpha) < 1e-6
            n_flip += 1
            continue
        _close(_f64(nxt[om.f64_sites, c]), qo, 1e-7, 1e-9)
        _close(lj[c], ljo, 1e-9, 1e-9)
    assert n_flip <= 1


@pytest.mark.parametrize("name", ["readme", "normal32", "refmodel8", "alldists"])
def test_find_reasonable_epsilon_injected(oracle, name):
    """Hoffman-Gelman Alg. 4 (hmc.rs:479-535): the doubling/halving search lands on the same
    power of two as the oracle for every chain."""
    cp, om = _pair(oracle, name)
    C, rng = 80, np.random.default_rng(31)
    cells = f64_values_for(om, rng, C)
    p0 = rng.standard_normal((cp.d, C))
    eng = E.Engine(cp, C, seed=1)
    eng.set_values(cells)
    eps = eng.hmc_find_eps_injected(E.hmc_config(), p0)
    mism = 0
    for c in range(C):
        q = _f64(cells[om.f64_sites, c])
        lj0 = om.log_joint_at(cells[:, c], q)
        oe = om.find_reasonable_epsilon(cells[:, c], q, lj0, p0[:, c])
        mism += int(eps[c] != oe)
    assert mism <= 1, mism        # a log-ratio within 1e-9 of ln 0.5 / ln 2 may tip the other way


@pytest.mark.parametrize("name,mode", [("readme", E.GRAD_FD_DENSE), ("normal32", E.GRAD_FD_DENSE),
                                       ("normal32", E.GRAD_FD_SPARSE), ("refmodel8", E.GRAD_FD_DENSE),
                                       ("ridge", E.GRAD_FD_SPARSE)])
def test_hmc_chain_matches_oracle(oracle, name, mode):
    """hmc_chain end to end (prior init, eps search, dual averaging, frozen sampling) vs the
    oracle run on the same Philox streams: every recorded draw within 1e-6 relative."""
    cp, om = _pair(oracle, name)
    C, nw, ns = 96, 30, 20
    cfg = E.hmc_config(grad_mode=mode, n_leapfrog=8)
    eng = E.Engine(cp, C, seed=5, chain_offset=7)
    d_draws = eng.device_alloc(ns * cp.d * C * 8)
    st = eng.hmc_run(cfg, ns, nw, d_draws)
    draws = eng.download(d_draws, (ns, cp.d, C))
    eng.device_free(d_draws)
    ocfg = oracle.HmcConfig.default(n_leapfrog=8)
    odraws, ofinal, oeps, ost = om.hmc_run(5, C, nw, ns, ocfg, chain0=7, n_threads=8)
    bad = ~np.isclose(draws, odraws, rtol=1e-6, atol=1e-8)
    bad_chains = np.unique(np.nonzero(bad)[2])
    assert len(bad_chains) <= 1, (len(bad_chains), draws[bad][:4], odraws[bad][:4])
    _close(eng.hmc_step_sizes()[np.setdiff1d(np.arange(C), bad_chains)],
           oeps[np.setdiff1d(np.arange(C), bad_chains)], 1e-6)
    assert abs(st.accept_rate - ost.accept_rate) < 2e-3
    assert st.n_divergent == ost.n_divergent or len(bad_chains) > 0


def test_hmc_mass_adaptation_matches_oracle(oracle):
    """adapt_mass path (Welford, reset at n_warmup/2, second eps search; hmc.rs:882-908)."""
    prog = W.normal_sites(6)
    cp, om = E.compile_model(prog), oracle.OracleModel(prog)
    C, nw, ns = 64, 40, 10
    cfg = E.hmc_config(adapt_mass=True, n_leapfrog=6)
    eng = E.Engine(cp, C, seed=9)
    d_draws = eng.device_alloc(ns * cp.d * C * 8)
    eng.hmc_run(cfg, ns, nw, d_draws)
    draws = eng.download(d_draws, (ns, cp.d, C))
    ocfg = oracle.HmcConfig.default(adapt_mass=1, n_leapfrog=6)
    odraws, _, _, _ = om.hmc_run(9, C, nw, ns, ocfg, n_threads=8)
    bad_chains = np.unique(np.nonzero(~np.isclose(draws, odraws, rtol=1e-5, atol=1e-7))[2])
    assert len(bad_chains) <= 1, bad_chains


def test_hmc_posterior_closed_form():
    """BASELINE target: posterior mean within 1e-3 of the closed form on the 32-site Normal
    model (x#i ~ N(0.8 y_i, 0.2)); 16 384 chains x 200 draws here (the bench runs 65 536)."""
    prog = W.normal_sites(32)
    cp = E.compile_model(prog)
    C, nw, ns = 16384, 150, 200
    eng = E.Engine(cp, C, seed=1)
    d_draws = eng.device_alloc(ns * cp.d * C * 8)
    st = eng.hmc_run(E.hmc_config(grad_mode=E.GRAD_FD_SPARSE), ns, nw, d_draws)
    draws = eng.download(d_draws, (ns, cp.d, C))
    _, mean, var = W.normal_sites_truth(32)
    assert np.abs(draws.mean(axis=(0, 2)) - mean).max() < 1e-3
    assert np.abs(draws.var(axis=(0, 2)) - var).max() < 5e-3
    assert 0.6 < st.accept_rate < 0.95 and st.n_divergent == 0
