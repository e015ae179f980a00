"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs.  Integer / index results must match exactly; floating-point results
within the tolerances written beside each assertion."""
import numpy as np
from tests import knife
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


@pytest.mark.parametrize("name", ["readme", "normal32", "coin", "refmodel8", "ridge", "mixture", "alldists", "hier_scale", "ridge7", "linreg", "rand0", "rand1", "rand2", "rand3", "rand4", "rand5"])
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


@pytest.mark.parametrize("name", ["readme", "normal32", "refmodel8", "ridge", "alldists", "hier_scale", "mixture", "logistic", "poisson_glm", "hier_logsigma"])
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
            knife.used("test_hmc_transition_injected: accept decision", model=name, mode=mode, chain=c, u_minus_alpha=float(u[c] - oalpha))
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
        if eps[c] != oe:
            knife.used("test_find_reasonable_epsilon_injected: step size", model=name, chain=c, gpu=float(eps[c]), oracle=float(oe))
        mism += int(eps[c] != oe)
    assert mism <= 1, mism        # a log-ratio within 1e-9 of ln 0.5 / ln 2 may tip the other way


def _replay_chain(oracle, om, cp, seed, chain, cells0, pos, info, cfg_L, nw, target=0.8, mass_at=None):
    """Teacher-forced replay of one chain: every GPU transition is re-done by the oracle from the
    GPU's own previous state with the GPU's own step size, so ulp-level differences cannot
    compound through the (chaotic) step-size feedback.  Returns the number of knife-edge flips."""
    d = cp.d
    cells = cells0.copy()
    q = _f64(cells[om.f64_sites]).copy()
    lj = om.log_joint_at(cells, q)
    m_inv = np.ones(d)
    flips = 0
    for t in range(pos.shape[0]):
        if mass_at is not None and t == mass_at:         # hmc.rs:885-908
            m_inv = np.var(pos[:t, :, chain], axis=0, ddof=1)
            m_inv = np.where(np.isfinite(m_inv) & (m_inv > 1e-8), m_inv, 1.0)
            p0e, _ = oracle.hmc_momentum(seed, chain, 1, d, np.sqrt(1.0 / m_inv), purpose=3)
            eps_reset = om.find_reasonable_epsilon(cells, q, lj, p0e, m_inv=m_inv)
            assert info["step_size"][t, chain] == pytest.approx(eps_reset, rel=1e-12)
        eps = info["step_size"][t, chain]
        p0, u = oracle.hmc_momentum(seed, chain, t, d, np.sqrt(1.0 / m_inv))
        qo, ljo, oacc, oalpha, odiv = om.hmc_transition(cells, q, lj, eps, cfg_L, p0, u, m_inv=m_inv)
        assert odiv == bool(info["divergent"][t, chain])
        if not odiv:
            _close(info["accept_prob"][t, chain], oalpha, 1e-6, 1e-9)
            if oacc != bool(info["accepted"][t, chain]):
                assert abs(u - oalpha) < 1e-6
                knife.used("teacher-forced replay: accept decision", chain=chain, transition=t, u_minus_alpha=float(u - oalpha))
                flips += 1
            else:
                # the force is a central difference with h = 1e-5: one ulp of log pi (|lj| ~ 1e2 -> 1e-14) becomes
                # 1e-14 / 2h ~ 1e-9 of force noise per evaluation, ~1e-8 of position after L steps
                _close(pos[t, :, chain], qo, 1e-6, 1e-7)
        q = pos[t, :, chain].copy()                        # teacher forcing: continue from the GPU state
        for k, j in enumerate(om.f64_sites):
            cells[j] = q[k:k + 1].view(np.int64)[0]
        lj = om.log_joint_at(cells, q)
    return flips


@pytest.mark.parametrize("name,mode", [("readme", E.GRAD_FD_DENSE), ("normal32", E.GRAD_FD_DENSE),
                                       ("normal32", E.GRAD_FD_SPARSE), ("refmodel8", E.GRAD_FD_DENSE),
                                       ("ridge", E.GRAD_FD_SPARSE), ("alldists", E.GRAD_FD_DENSE),
                                       ("hier", E.GRAD_FD_SPARSE), ("hier", E.GRAD_FD_DENSE), ("ridge7", E.GRAD_FD_SPARSE),
                                       ("ridge8", E.GRAD_FD_SPARSE), ("ridge16", E.GRAD_FD_SPARSE), ("ridge32", E.GRAD_FD_SPARSE),
                                       ("hier_scale", E.GRAD_FD_SPARSE), ("linreg", E.GRAD_FD_SPARSE), ("mixture", E.GRAD_FD_SPARSE),
                                       ("rand0", E.GRAD_FD_SPARSE), ("rand1", E.GRAD_FD_DENSE), ("rand2", E.GRAD_FD_SPARSE), ("rand4", E.GRAD_FD_SPARSE)])
def test_hmc_session_matches_oracle_teacher_forced(oracle, name, mode):
    """HmcSession (hmc.rs:667-920) step by step: prior init, Alg. 4 step size, every transition's
    HmcStepInfo, the dual-averaging recursion and the frozen step size -- each checked against
    the oracle given the GPU's own previous state."""
    cp, om = _pair(oracle, name)
    C, nw, n, L, seed, c0 = 64, 14, 24, 6, 5, 7
    cfg = E.hmc_config(grad_mode=mode, n_leapfrog=L)
    eng = E.Engine(cp, C, seed=seed, chain_offset=c0)
    eng.hmc_init(cfg, nw)
    cells0 = eng.get_values()
    eps0 = eng.hmc_step_sizes()
    lj0 = eng.hmc_log_joint()
    pos, info = eng.hmc_step_info(n)
    flips = 0
    for c in range(C):
        ocells, oacc, _ = om.run_prior(seed, c0 + c)                        # HmcSession::new, hmc.rs:673-687
        for j in range(cp.S):
            if cp.site_vtypes[j] == 0:
                _close(_f64(cells0[j:j + 1, c]), _f64(ocells[j:j + 1]), 1e-11, 1e-300)
            else:
                assert cells0[j, c] == ocells[j]
        _close(lj0[c], oacc.sum(), 1e-9, 1e-9)
        q0 = _f64(cells0[om.f64_sites, c])
        p0e, _ = oracle.hmc_momentum(seed, c0 + c, 0, cp.d, purpose=3)
        assert eps0[c] == om.find_reasonable_epsilon(cells0[:, c], q0, om.log_joint_at(cells0[:, c], q0), p0e)
        # DualAveraging (hmc.rs:141-184) fed with the GPU's acceptance probabilities
        _, tr, frozen = oracle.dual_averaging(eps0[c], 0.8, info["accept_prob"][:nw, c])
        assert info["step_size"][0, c] == eps0[c]
        _close(info["step_size"][1:nw, c], tr[:nw - 1], 1e-12)
        _close(info["step_size"][nw:, c], np.full(n - nw, frozen), 1e-12)    # frozen kernel: hmc.rs:789-798
    for c in range(C):
        f = _replay_chain(oracle, om, cp, seed, c, cells0[:, c], pos, info, L, nw) if c0 == 0 else \
            _replay_chain_offset(oracle, om, cp, seed, c0, c, cells0[:, c], pos, info, L, nw)
        flips += f
    assert flips <= 1
    _close(eng.hmc_step_sizes(), info["step_size"][-1], 0.0)


def _replay_chain_offset(oracle, om, cp, seed, c0, c, cells0, pos, info, L, nw, mass_at=None):
    """_replay_chain for an engine whose chain 0 has global id c0 (RNG streams use global ids)."""
    sub_pos, sub_info = pos[:, :, c:c + 1], {k: v[:, c:c + 1] for k, v in info.items()}

    class _Shift:
        def __getattr__(self, k):
            return getattr(oracle, k)

        @staticmethod
        def hmc_momentum(seed_, chain, it, d, mass_sqrt=None, purpose=2):
            return oracle.hmc_momentum(seed_, c0 + c, it, d, mass_sqrt, purpose)
    return _replay_chain(_Shift(), om, cp, seed, 0, cells0, sub_pos, sub_info, L, nw, mass_at=mass_at)


def test_hmc_free_running_short_chain_matches_oracle(oracle):
    """hmc_chain end to end without teacher forcing, kept short: ocml-vs-glibc ulp differences
    enter the finite-difference force (x 1/2h = 5e4) and are amplified several-fold per
    transition by the step-size feedback (measured on normal32: 3e-10 after 1 transition, chains
    fully decorrelated after 25 adaptive ones), so only the first transitions can be compared
    draw for draw; the long adaptive run is covered by the teacher-forced test above."""
    cp, om = _pair(oracle, "normal32")
    C, nw, ns = 96, 3, 3
    eng = E.Engine(cp, C, seed=5, chain_offset=7)
    d_draws = eng.device_alloc(ns * cp.d * C * 8)
    st = eng.hmc_run(E.hmc_config(n_leapfrog=8), ns, nw, d_draws)
    draws = eng.download(d_draws, (ns, cp.d, C))
    eng.device_free(d_draws)
    odraws, _, oeps, ost = om.hmc_run(5, C, nw, ns, oracle.HmcConfig.default(n_leapfrog=8), chain0=7, n_threads=8)
    bad_chains = np.unique(np.nonzero(~np.isclose(draws, odraws, rtol=1e-4, atol=1e-5))[2])
    if len(bad_chains): knife.used("free-running adaptive hmc_chain: draws", chains=bad_chains.tolist(), max_abs_diff=float(np.abs(draws - odraws).max()))
    assert len(bad_chains) <= 1, (bad_chains, np.abs(draws - odraws).max())
    assert abs(st.accept_rate - ost.accept_rate) < 1e-3


@pytest.mark.parametrize("name", ["readme", "normal32"])
def test_hmc_fixed_step_long_chain_matches_oracle(oracle, name):
    """No adaptation (n_warmup = 0, pinned step size): 60 free-running transitions stay within 1e-6."""
    cp, om = _pair(oracle, name)
    C, ns = 64, 60
    cfg = E.hmc_config(n_leapfrog=8, init_step_size=0.15, grad_mode=E.GRAD_FD_SPARSE)
    eng = E.Engine(cp, C, seed=11)
    d_draws = eng.device_alloc(ns * cp.d * C * 8)
    eng.hmc_run(cfg, ns, 0, d_draws)
    draws = eng.download(d_draws, (ns, cp.d, C))
    odraws, _, _, _ = om.hmc_run(11, C, 0, ns, oracle.HmcConfig.default(n_leapfrog=8, init_step_size=0.15), n_threads=8)
    bad_chains = np.unique(np.nonzero(~np.isclose(draws, odraws, rtol=1e-6, atol=1e-8))[2])
    if len(bad_chains): knife.used("fixed-step free-running hmc_chain: draws", model=name, chains=bad_chains.tolist(), max_abs_diff=float(np.abs(draws - odraws).max()))
    assert len(bad_chains) <= 1, bad_chains


def test_hmc_mass_adaptation_teacher_forced(oracle):
    """adapt_mass path (Welford variances, reset at n_warmup/2, second eps search with the new
    mass, new DualAveraging; hmc.rs:882-908), replayed transition by transition."""
    prog = W.normal_sites(6)
    cp, om = E.compile_model(prog), oracle.OracleModel(prog)
    C, nw, n, L, seed = 64, 16, 22, 6, 9
    cfg = E.hmc_config(adapt_mass=True, n_leapfrog=L)
    eng = E.Engine(cp, C, seed=seed)
    eng.hmc_init(cfg, nw)
    cells0 = eng.get_values()
    pos, info = eng.hmc_step_info(n)
    m_inv = eng.hmc_mass()
    flips = 0
    for c in range(C):
        var = np.var(pos[:nw // 2, :, c], axis=0, ddof=1)
        _close(m_inv[:, c], np.where(var > 1e-8, var, 1.0), 1e-10)
        flips += _replay_chain(oracle, om, cp, seed, c, cells0[:, c], pos, info, L, nw, mass_at=nw // 2)
        # second dual-averaging run restarts from the re-tuned step size
        e_reset = info["step_size"][nw // 2, c]
        _, tr, frozen = oracle.dual_averaging(e_reset, 0.8, info["accept_prob"][nw // 2:nw, c])
        _close(info["step_size"][nw // 2 + 1:nw, c], tr[:nw - nw // 2 - 1], 1e-12)
        _close(info["step_size"][nw:, c], np.full(n - nw, frozen), 1e-12)
    assert flips <= 1


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


@pytest.mark.parametrize("name,adapt_mass,mode", [("normal32", False, E.GRAD_FD_SPARSE), ("normal32", True, E.GRAD_FD_SPARSE),
                                                  ("hier", True, E.GRAD_FD_SPARSE), ("readme", False, E.GRAD_FD_SPARSE),
                                                  ("normal32", True, E.GRAD_FD_DENSE), ("hier", False, E.GRAD_FD_DENSE),
                                                  ("ridge7", True, E.GRAD_FD_SPARSE), ("ridge", False, E.GRAD_FD_SPARSE),     # linear-predictor records
                                                  ("hier_scale", True, E.GRAD_FD_SPARSE), ("linreg", False, E.GRAD_FD_SPARSE)])   # general records
def test_hmc_multiwave_is_bit_identical(name, adapt_mass, mode, monkeypatch):
    """k_hmc_stream_steps splits a tile's coordinates over 1 ... 16 waves; the per-coordinate operations and
    their order are the same, so draws, step sizes, mass matrix and log-joint must agree BIT FOR BIT."""
    monkeypatch.setenv("FG_JIT", "0")          # the hand-written kernels themselves; the compiled form is compared with them in tests/test_gpu_jit.py
    cp = E.compile_model(ZOO[name]())
    C, nw, ns = 192, 40, 25
    out = []
    for W in (1, 2, 4, 8, 16):
        monkeypatch.setenv("FG_HMC_WAVES", str(W))
        eng = E.Engine(cp, C, seed=21, chain_offset=5)
        d = eng.device_alloc(ns * cp.d * C * 8)
        st = eng.hmc_run(E.hmc_config(grad_mode=mode, n_leapfrog=7, adapt_mass=adapt_mass), ns, nw, d)
        draws = eng.download(d, (ns, cp.d, C))
        eng.device_free(d)
        out.append((draws, eng.hmc_step_sizes(), eng.hmc_log_joint(), eng.get_values(), st.accept_rate, st.n_divergent,
                    eng.hmc_mass() if adapt_mass else None))
    for o in out[1:]:
        for a, b in zip(out[0], o):
            assert (a is None and b is None) or np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
    assert np.isfinite(out[0][0]).all() and 0.5 < out[0][4] <= 1.0


@pytest.mark.parametrize("name,adapt_mass,mode", [("alldists", True, E.GRAD_FD_SPARSE), ("alldists", False, E.GRAD_FD_DENSE),
                                                  ("poisson_glm", True, E.GRAD_FD_SPARSE), ("hier_logsigma", False, E.GRAD_FD_SPARSE), ("logistic", True, E.GRAD_FD_SPARSE),
                                                  ("hier_logsigma", True, E.GRAD_FD_DENSE), ("mixture", False, E.GRAD_FD_DENSE),
                                                  ("hier_scale", True, E.GRAD_FD_DENSE), ("rand3", False, E.GRAD_FD_DENSE)])
def test_hmc_interp_multiwave_is_bit_identical(name, adapt_mass, mode, monkeypatch):
    """Programs without a gradient stream (expression parameters, selects, guards ...) share a tile between W waves, each on its
    own copy of the slots (k_hmc_interp_mw_steps, fg_hmc_interp.hip): per coordinate the same operations in the same order as
    the one-wave kernel, so draws, step sizes, mass matrix, log-joint and values agree BIT FOR BIT for every W."""
    cp = E.compile_model(ZOO[name]())
    if cp.d < 2:
        pytest.skip("one coordinate: nothing to split")
    C, nw, ns = 150, 30, 20
    out, kernels = [], []
    monkeypatch.setenv("FG_JIT", "0")                                      # the interpreter kernels themselves (tests/test_gpu_jit.py covers the compiled form)
    for mw, W, occ, pl in [(0, 0, 4, 1), (1, 2, 4, 1), (1, 3, 2, 1), (1, 4, 4, 0), (1, 8, 2, 0), (1, 12, 4, 1), (1, 16, 4, 1), (1, 0, 4, 1)]:
        monkeypatch.setenv("FG_HMC_INTERP_MW", str(mw))
        monkeypatch.setenv("FG_HMC_INTERP_OCC", str(occ))
        monkeypatch.setenv("FG_HMC_INTERP_LDSPROG", str(pl))           # sub-programs staged in LDS / fetched from global memory
        if W: monkeypatch.setenv("FG_HMC_INTERP_WAVES", str(W))
        else: monkeypatch.delenv("FG_HMC_INTERP_WAVES", raising=False)
        eng = E.Engine(cp, C, seed=23, chain_offset=9)
        d = eng.device_alloc(ns * cp.d * C * 8)
        st = eng.hmc_run(E.hmc_config(grad_mode=mode, n_leapfrog=5, adapt_mass=adapt_mass), ns, nw, d)
        kernels.append(eng.hmc_last_kernel())
        draws = eng.download(d, (ns, cp.d, C))
        eng.device_free(d)
        out.append((draws, eng.hmc_step_sizes(), eng.hmc_log_joint(), eng.get_values(), st.accept_rate, st.n_divergent,
                    eng.hmc_mass() if adapt_mass else None))
    assert kernels[0] == "k_hmc_steps W=1" and all(k.startswith("k_hmc_interp_mw_steps W=") for k in kernels[1:]), kernels
    for o in out[1:]:
        for a, b in zip(out[0], o):
            assert (a is None and b is None) or np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
    assert np.isfinite(out[0][0]).all()


@pytest.mark.parametrize("mode", [E.GRAD_FD_SPARSE, E.GRAD_FD_DENSE, E.GRAD_ANALYTIC])
@pytest.mark.parametrize("name,adapt_mass", [("normal32", False), ("normal32", True), ("readme", False), ("indep_mixed", True), ("indep_mixed", False),
                                             ("indep_uniform5", True)])
def test_hmc_sep_kernel_is_bit_identical(name, adapt_mass, mode, monkeypatch):
    """Independent-sites programs run whole trajectories in registers (k_hmc_sep_steps, 1..16 waves per tile) and evaluate
    the endpoint score as parallel terms summed in program order; the arithmetic per coordinate and per accumulator is the
    gradient-stream kernel's, so draws, step sizes, mass matrix, log-joint and statistics agree BIT FOR BIT with it -- in the
    dependency-aware mode and in the dense mode (grad_log_joint verbatim: whole log-joints at q +- h e_i, formed from term rows with
    the coordinate's own terms substituted, against the dense stream that re-evaluates every statement)."""
    monkeypatch.setenv("FG_JIT", "0")          # the hand-written kernels themselves; the compiled form is compared with them in tests/test_gpu_jit.py
    cp = E.compile_model(ZOO[name]())
    assert cp.stream_records[0] > 0 and lib_sep_records(cp) > 0
    C, nw, ns = 150, 40, 25
    out = []
    # half tiles (32 chains per workgroup, a Box-Muller pair's two coordinates in the two lane halves): sparse mode, programs whose
    # coordinates share one record shape with power-of-two sigmas -- elsewhere the switch is ignored and the run repeats the full tile
    layouts = [(0, 1, 0), (1, 1, 0), (1, 2, 0), (1, 4, 0), (1, 8, 0), (1, 16, 0)]
    if mode == E.GRAD_FD_SPARSE and name in ("normal32", "indep_uniform5"):
        layouts += [(1, 1, 1), (1, 2, 1), (1, 4, 1), (1, 16, 1), (1, 1, 2), (1, 2, 2), (1, 8, 2)]      # 2: quarter tiles (16 chains, four coordinates per wave)
    layouts = [lay + (1,) for lay in layouts]
    if mode == E.GRAD_FD_DENSE:               # the dense mode's two forms: coordinates in LDS rows (any program) and in registers (one shape of record, powers of two)
        layouts += [(1, W, 0, 0) for W in (1, 4, 8, 16)]
    for sep, W, half, dense_regs in layouts:
        monkeypatch.setenv("FG_HMC_SEP", str(sep))
        monkeypatch.setenv("FG_HMC_WAVES", str(W))
        monkeypatch.setenv("FG_HMC_SEP_HALF", str(half))
        monkeypatch.setenv("FG_HMC_DENSE_FAST", str(dense_regs))
        eng = E.Engine(cp, C, seed=21, chain_offset=5)
        d = eng.device_alloc(ns * cp.d * C * 8)
        st = eng.hmc_run(E.hmc_config(grad_mode=mode, n_leapfrog=7, adapt_mass=adapt_mass), ns, nw, d)
        draws = eng.download(d, (ns, cp.d, C))
        eng.device_free(d)
        pos, info = eng.hmc_step_info(3)
        assert ("half tiles" in eng.hmc_last_kernel()) == (half == 1) and ("quarter tiles" in eng.hmc_last_kernel()) == (half == 2)
        assert ("k_hmc_sep_steps" in eng.hmc_last_kernel()) == bool(sep)
        if name == "normal32":
            assert ("coordinates in registers" in eng.hmc_last_kernel()) == bool(sep and mode == E.GRAD_FD_DENSE and dense_regs)
        out.append((draws, eng.hmc_step_sizes(), eng.hmc_log_joint(), eng.get_values(), st.accept_rate, st.n_divergent,
                    eng.hmc_mass() if adapt_mass else None, pos, info["accept_prob"], info["accepted"], info["step_size"]))
        eng.close()
    for o in out[1:]:
        for a, b in zip(out[0], o):
            assert (a is None and b is None) or np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
    assert np.isfinite(out[0][0]).all() and 0.5 < out[0][4] <= 1.0


@pytest.mark.parametrize("name,adapt_mass", [("ridge8", False), ("ridge8", True), ("ridge16", False), ("ridge16", True), ("ridge32", False), ("ridge32", True),
                                             ("ridge24", True), ("ridge12", False), ("ridge7", True), ("ridge5", False), ("ridge64", True), ("ridge40", False)])
def test_hmc_lin_kernel_is_bit_identical(name, adapt_mass, monkeypatch):
    """Dense regressions take the observation-major gradient (k_hmc_lin_steps: a wave owns D / W term positions, forms every
    observation's products and prefix sums once and carries the suffix sums of its own coordinates side by side).  Per
    (coordinate, observation, sign) the additions and multiplications are the gradient stream's in the same order, so draws,
    step sizes, mass matrix, log-joint, statistics and per-transition info agree BIT FOR BIT with k_hmc_stream_steps -- for
    both waves-per-tile layouts, on full (64-chain) and half (32-chain) tiles.  The kernel is built for 8, 16 and 32 term positions:
    any other coefficient count up to 32 runs padded with terms that read the always-zero slot (24, 12, 7, 5 coefficients here); 33 .. 64
    coefficients take the 64-position build (64, 40 here)."""
    monkeypatch.setenv("FG_JIT", "0")          # the hand-written kernels themselves; the compiled form is compared with them in tests/test_gpu_jit.py
    cp = E.compile_model(ZOO[name]())
    assert E.lib().fg_program_stream_records(cp.h, 4) > 0
    C, nw, ns = 150, 40, 25
    out = []
    dp = 8 if cp.d <= 8 else (16 if cp.d <= 16 else (32 if cp.d <= 32 else 64))
    layouts = ((0, 1, 0), (1, dp // 4, 0), (1, dp // 2, 0), (1, dp // 4, 1), (1, dp // 2, 1))
    if dp == 64: layouts = ((0, 1, 0), (1, 16, 0), (1, 8, 0))   # 33 .. 64 coefficients: q read from LDS a chunk at a time; full tiles
    elif dp >= 16: layouts += ((1, dp // 8, 0), (1, dp // 8, 1))   # eight positions per wave (256 VGPRs, two waves per SIMD): the default since round 4
    for lin, W_, half in layouts:
        monkeypatch.setenv("FG_HMC_LIN", str(lin))
        monkeypatch.setenv("FG_HMC_WAVES", str(W_))
        monkeypatch.setenv("FG_HMC_LIN_HALF", str(half))      # half tiles: 32 chains per workgroup, the two signs of the finite difference in the two lane halves
        eng = E.Engine(cp, C, seed=21, chain_offset=5)
        d = eng.device_alloc(ns * cp.d * C * 8)
        st = eng.hmc_run(E.hmc_config(n_leapfrog=7, adapt_mass=adapt_mass), ns, nw, d)
        draws = eng.download(d, (ns, cp.d, C))
        eng.device_free(d)
        pos, info = eng.hmc_step_info(3)
        assert ("k_hmc_lin_steps" in eng.hmc_last_kernel()) == bool(lin), eng.hmc_last_kernel()
        if lin: assert eng.hmc_last_kernel().endswith(f"W={W_}"), eng.hmc_last_kernel()
        out.append((draws, eng.hmc_step_sizes(), eng.hmc_log_joint(), eng.get_values(), st.accept_rate, st.n_divergent,
                    eng.hmc_mass() if adapt_mass else None, pos, info["accept_prob"], info["accepted"], info["step_size"]))
        eng.close()
    for o in out[1:]:
        for a, b in zip(out[0], o):
            assert (a is None and b is None) or np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
    assert np.isfinite(out[0][0]).all() and 0.5 < out[0][4] <= 1.0


def test_hmc_lin_kernel_with_huge_positions_is_bit_identical(monkeypatch):
    """The eight-coordinate layout forms -0.5 z z - ln sigma as one fma (exact product) and re-runs a wave's gradient through the unfused
    instance when a force component comes out non-finite -- the one place the two forms can differ is z z in [2^1024, 2^1025).  Chains started
    from |beta| ~ 1e152 ... 1e200 (standardised residuals around and beyond that window, sums that overflow, NaN differences) must leave the
    observation-major kernel in exactly the state the gradient stream leaves them in: values, step sizes, log-joint, divergence counts."""
    monkeypatch.setenv("FG_JIT", "0")
    cp = E.compile_model(ZOO["ridge32"]())
    C = 128
    out = []
    for lin, half in ((0, 0), (1, 0), (1, 1)):
        monkeypatch.setenv("FG_HMC_LIN", str(lin))
        monkeypatch.setenv("FG_HMC_LIN_HALF", str(half))
        eng = E.Engine(cp, C, seed=11)
        eng.prior_init(0)
        cells = _f64(eng.get_values()).copy()
        rs = np.random.RandomState(5)
        for c in range(0, C, 2):                                   # every other chain: one or a few coefficients far out
            for j in rs.choice(cp.S, size=1 + (c // 2) % 3, replace=False):
                cells[j, c] = rs.choice([-1.0, 1.0]) * 10.0 ** rs.uniform(152.0, 156.0 if c % 4 == 0 else 200.0)
        eng.hmc_init(E.hmc_config(n_leapfrog=5, init_step_size=1e-3), 0)      # (draws the initial state from the prior)
        eng.set_values(cells.view(np.int64))                                   # ... replaced here; the log-joint is re-scored
        eng.hmc_step(4)
        if lin: assert "k_hmc_lin_steps" in eng.hmc_last_kernel() and ("half" in eng.hmc_last_kernel()) == bool(half), eng.hmc_last_kernel()
        st = eng.hmc_stats()
        out.append((eng.get_values(), eng.hmc_step_sizes(), eng.hmc_log_joint(), st.n_divergent, st.accept_rate))
        eng.close()
    assert out[0][3] > 0                                           # the far-out chains do diverge
    for o in out[1:]:
        for a, b in zip(out[0], o):
            assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def lib_sep_records(cp):
    return E.lib().fg_program_stream_records(cp.h, 3)


def test_hmc_config_is_validated_and_tiny_engines_run():
    """n_leapfrog = 0 (the reference's loop then never moves, hmc.rs:385) and a non-positive finite-difference step are
    refused at the boundary; one chain, and a chain count that leaves most lanes of the last tile empty, run."""
    cp = E.compile_model(ZOO["normal32"]())
    eng = E.Engine(cp, 1, seed=4)
    for bad in (E.hmc_config(n_leapfrog=0), E.hmc_config(finite_diff_eps=0.0)):
        with pytest.raises(E.EngineError) as ei:
            eng.hmc_init(bad, 0)
        assert ei.value.code == E.FG_E_BAD_ARG
    for C in (1, 65):
        eng = E.Engine(cp, C, seed=4)
        d = eng.device_alloc(5 * cp.d * C * 8)
        st = eng.hmc_run(E.hmc_config(grad_mode=E.GRAD_FD_SPARSE), 5, 10, d)
        x = eng.download(d, (5, cp.d, C))
        assert np.isfinite(x).all() and st.n_divergent == 0
        eng.device_free(d)
        ref = E.Engine(cp, 130, seed=4)                               # the same chains inside a larger engine: identical draws
        d2 = ref.device_alloc(5 * cp.d * 130 * 8)
        ref.hmc_run(E.hmc_config(grad_mode=E.GRAD_FD_SPARSE), 5, 10, d2)
        assert np.array_equal(ref.download(d2, (5, cp.d, 130))[:, :, :C], x)


@pytest.mark.parametrize("name", ["readme", "normal32", "refmodel8", "hier", "ridge", "ridge7", "linreg"])
def test_analytic_gradient_matches_finite_difference(oracle, name):
    """FG_GRAD_ANALYTIC (closed form for Normal / linear-predictor force terms) against the oracle's central difference
    (hmc.rs:304-329): they differ by the finite difference's own O(h^2) + rounding error -- the same 5e-6 * (1 + |lj|)
    bound the FD parity test uses -- and against a numpy derivative of the same density where that is easy."""
    cp, om = _pair(oracle, name)
    C = 70
    cells = f64_values_for(om, np.random.default_rng(11), C)
    eng = E.Engine(cp, C, seed=1)
    eng.set_values(cells)
    g_an, ok = eng.hmc_grad(1e-5, E.GRAD_ANALYTIC)
    g_fd, _ = eng.hmc_grad(1e-5, E.GRAD_FD_SPARSE)
    assert ok.all() and np.isfinite(g_an).all()
    for c in range(0, C, 3):
        q = _f64(cells[om.f64_sites, c])
        og, _ = om.grad_log_joint(cells[:, c], q)
        tol = 5e-6 * (1.0 + abs(om.log_joint_at(cells[:, c], q)))
        _close(g_an[:, c], og, 1e-7, tol)
    _close(g_an, g_fd, 1e-6, 1e-4)
    if name == "normal32":                                # d/dx [-x^2/2 - (y - x)^2 / (2 * 0.25)]
        x = _f64(cells)
        y = np.array([0.2 * int(n.split("#")[1]) - 1.0 for n in cp.site_names])[:, None]
        _close(g_an, -x + (y - x) / 0.25, 1e-13, 1e-13)


def test_analytic_mode_is_refused_where_unavailable(monkeypatch):
    """Force terms other than Normals with constant sigma take their analytic gradient from the unit compiled at run time (round 4,
    tests/test_gpu_jit.py); without the run-time compiler there is none and the mode is refused -- never a silent finite difference."""
    monkeypatch.setenv("FG_JIT", "0")
    for name in ("hier_scale", "mixture", "alldists"):
        eng = E.Engine(E.compile_model(ZOO[name]()), 64, seed=1)
        with pytest.raises(E.EngineError) as ei:
            eng.hmc_init(E.hmc_config(grad_mode=E.GRAD_ANALYTIC), 10)
        assert ei.value.code == E.FG_E_UNSUPPORTED
        eng.close()
    monkeypatch.delenv("FG_JIT")
    eng = E.Engine(E.compile_model(ZOO["hier_scale"]()), 64, seed=1)
    eng.hmc_init(E.hmc_config(grad_mode=E.GRAD_ANALYTIC), 10)
    eng.hmc_step(3)
    assert eng.hmc_last_kernel().startswith("k_hmc_jit_steps")
    eng.close()


def test_issue_priorities_change_no_result(monkeypatch):
    """s_setprio only reorders the waves of a SIMD: the register-resident HMC kernel (waves taking turns, 8 waves per tile) and the
    multi-wave MH kernel (control wave, phase B) give the same bits with the priorities switched off."""
    monkeypatch.setenv("FG_JIT", "0")          # the hand-written kernels themselves; the compiled form is compared with them in tests/test_gpu_jit.py
    out = {}
    for prio in ("1", "0"):
        monkeypatch.setenv("FG_HMC_PRIO", prio)
        monkeypatch.setenv("FG_MH_PRIO", prio)
        monkeypatch.setenv("FG_HMC_WAVES", "8")
        res = []
        for name in ("normal32", "refmodel8"):
            cp = E.compile_model(ZOO[name]())
            C, nw, ns = 320, 20, 15
            eng = E.Engine(cp, C, seed=6)
            d = eng.device_alloc(ns * cp.d * C * 8)
            eng.hmc_run(E.hmc_config(n_leapfrog=5), ns, nw, d)
            res.append(eng.download(d, (ns, cp.d, C)))
            eng.device_free(d)
            eng.close()
            eng = E.Engine(cp, C, seed=6)
            eng.mh_init(40); eng.mh_step(80); eng.synchronize()
            res.append(eng.get_values()); res.append(eng.mh_scales())
            eng.close()
        out[prio] = res
    for a, b in zip(out["1"], out["0"]):
        assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.mark.parametrize("name", ["normal32", "ridge"])
def test_hmc_analytic_mode_posterior_and_wave_invariance(name, monkeypatch):
    """hmc_chain with FG_GRAD_ANALYTIC: bit-identical for 1, 2, 4 waves per tile, close to the finite-difference chain
    for a few fixed-step transitions, and the closed-form posterior mean is recovered."""
    cp = E.compile_model(ZOO[name]())
    C, nw, ns = 256, 60, 60
    out = []
    for W_ in (1, 2, 4, 8):
        monkeypatch.setenv("FG_HMC_WAVES", str(W_))
        eng = E.Engine(cp, C, seed=8)
        d = eng.device_alloc(ns * cp.d * C * 8)
        st = eng.hmc_run(E.hmc_config(grad_mode=E.GRAD_ANALYTIC, n_leapfrog=8, adapt_mass=True), ns, nw, d)
        out.append((eng.download(d, (ns, cp.d, C)), eng.hmc_step_sizes(), st.accept_rate))
        eng.device_free(d)
    for o in out[1:]:
        assert np.array_equal(o[0], out[0][0]) and np.array_equal(o[1], out[0][1])
    assert 0.6 < out[0][2] <= 1.0
    monkeypatch.delenv("FG_HMC_WAVES")
    res = {}
    for mode in (E.GRAD_ANALYTIC, E.GRAD_FD_SPARSE):      # fixed step size: no chaotic step-size feedback
        eng = E.Engine(cp, C, seed=8)
        d = eng.device_alloc(4 * cp.d * C * 8)
        eng.hmc_run(E.hmc_config(grad_mode=mode, n_leapfrog=8, init_step_size=0.05), 4, 0, d)
        res[mode] = eng.download(d, (4, cp.d, C))
        eng.device_free(d)
    same = np.isclose(res[E.GRAD_ANALYTIC], res[E.GRAD_FD_SPARSE], rtol=1e-5, atol=1e-6).all(axis=(0, 1))
    assert same.sum() >= C - 2                             # an accept on a knife edge may flip a chain
    if name == "normal32":
        _, tm, _ = W.normal_sites_truth(32)
        assert np.abs(out[0][0].mean(axis=(0, 2)) - tm).max() < 0.02
