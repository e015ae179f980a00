"""GPU parity: adaptive single-site MH (adaptive_mcmc_chain, src/inference/mh.rs) through the
C ABI against the CPU oracle on the same Philox streams.  Discrete sites and accept decisions
must match exactly; f64 sites to 1e-9 (MH has no finite-difference amplification)."""
import numpy as np
import pytest

from fugue_amd import engine as E
from fugue_amd import model as M
from fugue_amd import workloads as W
from tests.models import ZOO
from tests.random_models import random_program

pytestmark = pytest.mark.gpu


def _run_both(oracle, prog, C, nw, ns, seed, overrides=None, chain0=0):
    cp, om = E.compile_model(prog), oracle.OracleModel(prog)
    eng = E.Engine(cp, C, seed=seed, chain_offset=chain0)
    rec = list(range(cp.S))
    d_draws = eng.device_alloc(max(1, ns * cp.S * C) * 8)
    st = eng.mh_run(ns, nw, overrides, rec, d_draws)
    draws = eng.download(d_draws, (ns, cp.S, C), dtype=np.int64)
    eng.device_free(d_draws)
    odraws, ofinal, oscales, ost = om.mh_run(seed, C, nw, ns, overrides, rec, chain0=chain0, n_threads=8)
    return cp, eng, st, draws, odraws, ofinal, oscales, ost


def _compare(cp, draws, odraws, max_bad_chains=0):
    bad = np.zeros(draws.shape[2], dtype=bool)
    for j in range(cp.S):
        if cp.site_vtypes[j] == 0:
            g, o = draws[:, j, :].view(np.float64), odraws[:, j, :].view(np.float64)
            bad |= (~np.isclose(g, o, rtol=1e-9, atol=1e-12)).any(axis=0)
        else:
            bad |= (draws[:, j, :] != odraws[:, j, :]).any(axis=0)
    if bad.any() and max_bad_chains > 0:
        from tests import knife
        knife.used("MH draws against the oracle", chains=np.nonzero(bad)[0][:10].tolist())
    assert bad.sum() <= max_bad_chains, np.nonzero(bad)[0][:10]
    return bad


@pytest.mark.parametrize("name", ["readme", "coin", "refmodel8", "mixture", "alldists", "normal32", "hier_scale", "ridge7", "linreg", "logistic", "poisson_glm", "hier_logsigma"])
def test_mh_chain_matches_oracle(oracle, name):
    prog = ZOO[name]()
    cp, eng, st, draws, odraws, ofinal, oscales, ost = _run_both(oracle, prog, C=96, nw=150, ns=60, seed=13, chain0=3)
    # a proposal whose acceptance sits on a 1e-13 knife edge may flip on one chain
    bad = _compare(cp, draws, odraws, max_bad_chains=1)
    assert abs(st.accept_rate - ost.accept_rate) < 2e-3
    same = np.isclose(eng.mh_scales(), oscales, rtol=1e-10).all(axis=0)
    assert same.sum() >= 95                                    # frozen after warmup: mh.rs:989-1001
    # final state of every chain whose recorded draws matched: discrete cells exact, f64 cells to the draws' tolerance
    good = ~bad
    final = eng.get_values()
    for j in range(cp.S):
        if cp.site_vtypes[j] == 0:
            assert np.allclose(final[j, good].view(np.float64), ofinal[j, good].view(np.float64), rtol=1e-9, atol=1e-12), cp.site_names[j]
        else:
            assert np.array_equal(final[j, good], ofinal[j, good]), cp.site_names[j]


@pytest.mark.parametrize("seed", range(12))
def test_mh_random_models_match_oracle(oracle, seed):
    """Random programs (tests/random_models.py: chains of Normals, scale / rate / probability sites, Categorical selects with zero
    entries in their tables, free discrete sites) through adaptive_mcmc_chain against the oracle: discrete draws exact, f64 1e-9."""
    prog = random_program(1000 + seed)
    cp, eng, st, draws, odraws, ofinal, oscales, ost = _run_both(oracle, prog, C=64, nw=80, ns=40, seed=21 + seed, chain0=seed)
    _compare(cp, draws, odraws, max_bad_chains=1)
    assert abs(st.accept_rate - ost.accept_rate) < 3e-3


def test_mh_positive_support_uses_log_space_walk(oracle):
    """Gamma(3,2): support-based kind detection picks the log-space walk with its Jacobian
    (mh.rs:339-358, 201-224); mean 1.5 (tests/f_mcmc_proposals.rs:31-70)."""
    prog = W.gamma_scale_model()
    cp, eng, st, draws, odraws, *_ = _run_both(oracle, prog, C=512, nw=300, ns=300, seed=2)
    _compare(cp, draws, odraws, max_bad_chains=2)
    x = draws[:, 0, :].view(np.float64)
    assert (x > 0).all()
    assert abs(x.mean() - 1.5) < 0.03


def test_mh_overrides(oracle):
    """adaptive_mcmc_chain_with_overrides (mh.rs:938-944): Reflect, PriorResample and a forced
    Gaussian walk on a positive-support site."""
    P = M.Program()
    a = P.sample(M.addr("a"), M.Uniform(0.0, 2.0))
    g = P.sample(M.addr("g"), M.Gamma(2.0, 1.0))
    t = P.sample(M.addr("t"), M.Normal(0.0, 2.0))
    P.observe(M.addr("y"), M.Normal(a + t, 0.5 + g), 1.0)
    cp = E.compile_model(P)
    ov = [None] * cp.S
    ov[cp.site_names.index("a")] = (E.PROP_REFLECT, 0.0, 2.0)
    ov[cp.site_names.index("g")] = (E.PROP_GAUSSIAN, 0.0, 0.0)
    ov[cp.site_names.index("t")] = (E.PROP_PRIOR_RESAMPLE, 0.0, 0.0)
    cp, eng, st, draws, odraws, *_ = _run_both(oracle, P, C=128, nw=100, ns=100, seed=4, overrides=ov)
    _compare(cp, draws, odraws, max_bad_chains=1)
    av = draws[:, cp.site_names.index("a"), :].view(np.float64)
    assert (av >= 0).all() and (av <= 2).all()


def test_mh_conjugate_posterior():
    """README model by MH: posterior N(0.96, 0.2), 4096 chains."""
    cp = E.compile_model(W.readme_normal())
    C, nw, ns = 4096, 500, 400
    eng = E.Engine(cp, C, seed=1)
    d = eng.device_alloc(ns * C * 8)
    st = eng.mh_run(ns, nw, None, [0], d)
    x = eng.download(d, (ns, 1, C))
    assert abs(x.mean() - 0.96) < 5e-3 and abs(x.var() - 0.2) < 5e-3
    assert 0.3 < st.accept_rate < 0.6                             # adapts towards 0.44 (mh.rs:946)


def test_mh_mixture_recovers_component_means():
    """C5 pattern (Categorical assignments + Normal means) at small scale: sorted posterior
    means of mu#k near the generating (-6,-2,2,6)."""
    data, _ = W.mixture_data(64)
    cp = E.compile_model(W.mixture(data))
    C, nw, ns = 256, 3000, 200
    eng = E.Engine(cp, C, seed=3)
    mu_sites = [cp.site_names.index(f"mu#{k}") for k in range(4)]
    d = eng.device_alloc(ns * 4 * C * 8)
    eng.mh_run(ns, nw, None, mu_sites, d)
    mu = eng.download(d, (ns, 4, C))
    srt = np.sort(mu, axis=1)
    med = np.median(srt.mean(axis=0), axis=1)                     # robust to a few label-merged chains
    assert np.abs(med - np.array([-6.0, -2.0, 2.0, 6.0])).max() < 1.0


@pytest.mark.parametrize("name", ["refmodel8", "normal32", "mixture", "hier_scale", "linreg", "readme", "coin"])
def test_mh_multiwave_kernel_is_identical_to_the_one_wave_kernel(name, monkeypatch):
    """k_mh_mw_steps (W waves per tile: parallel log-density terms, in-order sums, pipelined random numbers) reproduces
    k_mh_steps exactly: recorded draws, final state, adapted scales, log-weights and accept counts, for every W."""
    prog = ZOO[name]()
    cp = E.compile_model(prog)
    C, nw, ns = 150, 120, 40
    rec = list(range(cp.S))
    out = []
    monkeypatch.setenv("FG_JIT", "0")                                      # the hand-written phase B (the generated one: the next test)
    # (multi-wave kernel?, waves per tile, the two in-order sums on two waves? -- the engine picks that for programs of >= 64 statements)
    # pipe: the one-control-wave loop (the default) or the step's serial recipe split over waves (FG_MH_PIPE=1, fg_mh_mw2_body.h:
    # decider / speculative proposer -- kept as an opt-in: identical results, measured slower)
    kernels = []
    for mw, W, split, pipe in ((0, 1, 0, 1), (1, 2, 0, 1), (1, 4, 0, 1), (1, 8, 0, 1), (1, 16, 0, 1), (1, 2, 1, 1), (1, 3, 1, 1), (1, 4, 1, 1), (1, 16, 1, 1),
                               (1, 2, 0, 0), (1, 4, 0, 0), (1, 16, 0, 0), (1, 4, 1, 0), (1, 16, 1, 0)):
        monkeypatch.setenv("FG_MH_MW", str(mw))
        monkeypatch.setenv("FG_HMC_WAVES", str(W))
        monkeypatch.setenv("FG_MH_SPLIT", str(split))
        monkeypatch.setenv("FG_MH_PIPE", str(pipe))
        eng = E.Engine(cp, C, seed=13, chain_offset=3)
        d = eng.device_alloc(max(1, ns * cp.S * C) * 8)
        st = eng.mh_run(ns, nw, None, rec, d)
        draws = eng.download(d, (ns, cp.S, C), dtype=np.int64)
        eng.device_free(d)
        kernels.append(eng.mh_last_kernel())
        out.append((draws, eng.get_values(), eng.mh_scales(), eng.mh_log_weight(), st.accept_rate))
        eng.close()
    if name != "coin":                                                     # (an expression parameter: no score stream, the interpreter kernels)
        assert kernels[1].startswith("k_mh_mw2_steps") and kernels[-1].startswith("k_mh_mw_steps"), kernels
    for k, o in zip(kernels[1:], out[1:]):
        for a, b in zip(out[0], o):
            assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True), k


@pytest.mark.parametrize("name", ["hier_scale", "linreg", "ridge7", "hier_mixed", "refmodel20", "mixture", "readme"])
def test_mh_multiwave_kernel_with_compiled_statements_is_identical(name, monkeypatch):
    """The same kernel with the general records of its phase B (those fg_score_one evaluates; the operand-pattern runs of plain
    Normal records stay hand-written) generated from the program and compiled at run time (fg_jit_mhmw_source; sixteen statement
    segments dealt to the waves): the one-wave kernel's draws, state, scales, log-weights and accept counts, for every W.
    (Short programs of pattern records only kept the hand-written kernel until round 4's survey: profiles/round4_zoo_mh.txt.)"""
    if name == "hier_mixed":                            # pattern records (sigma = 1, 2) and general ones (sigma a site, sigma = 0.7) in one program
        prog = M.Program()
        tau = prog.sample(M.addr("tau"), M.Gamma(2.0, 1.5))
        for i in range(6):
            m = prog.sample(M.addr("m", i), M.Normal(0.0, 2.0 if i % 2 else 0.7))
            prog.observe(M.addr("y", i), M.Normal(m, 1.0 if i < 3 else tau), 0.3 * i - 1.0)
    else:
        prog = ZOO[name]() if name in ZOO else W.reference_model(20)
    cp = E.compile_model(prog)
    C, nw, ns = 150, 120, 40
    rec = list(range(cp.S))
    out, kernels = [], []
    for jit, mw, Wv, split, pipe in ((0, 0, 1, 0, 1), (1, 1, 2, 0, 1), (1, 1, 4, 1, 1), (1, 1, 8, 0, 1), (1, 1, 16, 1, 1), (1, 1, 0, -1, 1), (1, 1, 4, 0, 0), (1, 1, 16, 1, 0)):
        monkeypatch.setenv("FG_JIT", str(jit))
        monkeypatch.setenv("FG_MH_MW", str(mw))
        monkeypatch.setenv("FG_MH_PIPE", str(pipe))
        if Wv: monkeypatch.setenv("FG_HMC_WAVES", str(Wv))
        else: monkeypatch.delenv("FG_HMC_WAVES", raising=False)
        if split >= 0: monkeypatch.setenv("FG_MH_SPLIT", str(split))
        else: monkeypatch.delenv("FG_MH_SPLIT", raising=False)
        eng = E.Engine(cp, C, seed=13, chain_offset=3)
        d = eng.device_alloc(max(1, ns * cp.S * C) * 8)
        st = eng.mh_run(ns, nw, None, rec, d)
        draws = eng.download(d, (ns, cp.S, C), dtype=np.int64)
        eng.device_free(d)
        kernels.append(eng.mh_last_kernel())
        out.append((draws, eng.get_values(), eng.mh_scales(), eng.mh_log_weight(), st.accept_rate))
        eng.close()
    # (every program without lookup records has all its statements generated since round 4, the two-statement README model included)
    want = ("k_mh_mw2_jit_steps", "k_mh_mw_jit_steps")
    assert all(k.startswith(want[0]) for k in kernels[1:6]) and all(k.startswith(want[1]) for k in kernels[6:]), kernels
    for o in out[1:]:
        for a, b in zip(out[0], o):
            assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.mark.parametrize("K,n", [(4, 10), (3, 21)])
def test_mh_mixture_without_rows_for_uniform_categorical_priors(K, n, monkeypatch):
    """The mixture pattern (examples/mixture_models.rs:77-112 with fixed equal weights): every assignment's log_prior term is the
    same constant ln(1/K), so the multi-wave kernel keeps no rows for them and the control wave adds the constants after the rows of
    log_prior, in order (FgMhSeg).  Against the one-wave kernel and against the kernel WITH those rows (FG_MH_CATU=0), also with
    indices injected out of range (their terms are -inf)."""
    data, _ = W.mixture_data(n)
    cp = E.compile_model(W.mixture(data, K=K))
    C, nw, ns = 200, 150, 60
    rec = list(range(cp.S))
    zs = [j for j in range(cp.S) if cp.site_vtypes[j] == 3]
    out = []
    for inject in (False, True):
        res = []
        for mw, catu, jit, Wv in ((0, 1, 0, 1), (1, 0, 0, 4), (1, 1, 0, 2), (1, 1, 0, 8), (1, 1, 1, 0), (1, 1, 1, 16)):
            monkeypatch.setenv("FG_MH_MW", str(mw)); monkeypatch.setenv("FG_MH_CATU", str(catu)); monkeypatch.setenv("FG_JIT", str(jit))
            if Wv: monkeypatch.setenv("FG_HMC_WAVES", str(Wv))
            else: monkeypatch.delenv("FG_HMC_WAVES", raising=False)
            eng = E.Engine(cp, C, seed=21, chain_offset=2)
            eng.mh_init(nw)                                         # draws the start from the prior (mh.rs:950-957)
            if inject:
                v = eng.get_values()
                v[zs[0], ::3] = K + 2; v[zs[1], 1::4] = -1; v[zs[-1], ::5] = K      # out of range in some chains, two sites at once in a few
                eng.set_values(v)                                   # (re-scores the cached log-weight: -inf in those chains)
                assert np.isneginf(eng.mh_log_weight()[::3]).all()
            eng.mh_step(nw)
            d = eng.device_alloc(ns * cp.S * C * 8)
            eng.mh_step(ns, rec, d)
            draws = eng.download(d, (ns, cp.S, C), dtype=np.int64)
            eng.device_free(d)
            res.append((draws, eng.get_values(), eng.mh_scales(), eng.mh_log_weight(), eng.mh_stats().accept_rate))
            eng.close()
        for o in res[1:]:
            for a, b in zip(res[0], o):
                assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
        out.append(res[0])
    assert not np.array_equal(out[0][0], out[1][0])                # the injected indices changed the chains
    # ... and such a chain stays where it is, as in the reference: log q(x|x') of the move back is -inf too, so log_alpha is NaN (mh.rs:731-733)
    assert np.isneginf(out[1][3][::3]).all() and np.isfinite(out[0][3]).all()


@pytest.mark.parametrize("name,with_overrides", [("alldists", False), ("alldists", True), ("poisson_glm", False), ("hier_logsigma", True), ("coin", False), ("logistic", False)])
def test_mh_interp_multiwave_is_bit_identical(name, with_overrides, monkeypatch):
    """Programs without a score stream split a step's scoring run between W waves (k_mh_interp_mw_steps, fg_mh_interp.hip: each
    statement's term in an LDS row, the three accumulators added in program order by wave 0); steps whose proposal needs the model
    (undecided kinds during the first steps, PriorResample overrides) run k_mh_steps' propose-and-score path on wave 0.  Recorded
    draws, final state, adapted scales, log-weights and accept counts equal the one-wave kernel's for every W."""
    cp = E.compile_model(ZOO[name]())
    C, nw, ns = 150, 100, 40
    rec = list(range(cp.S))
    ov = None
    if with_overrides:
        ov = [None] * cp.S
        f64 = [j for j in range(cp.S) if cp.site_vtypes[j] == 0]
        ov[f64[0]] = (E.PROP_PRIOR_RESAMPLE, 0.0, 0.0)           # needs the model at that site: the general path whenever a lane picks it
        ov[f64[1]] = (E.PROP_GAUSSIAN, 0.0, 0.0)
    out = []
    monkeypatch.setenv("FG_JIT", "0")                                      # the interpreter kernels themselves (tests/test_gpu_jit.py covers the compiled form)
    for mw, W in ((0, 0), (1, 2), (1, 3), (1, 4), (1, 8), (1, 0)):
        monkeypatch.setenv("FG_HMC_INTERP_MW", str(mw))
        if W: monkeypatch.setenv("FG_MH_INTERP_WAVES", str(W))
        else: monkeypatch.delenv("FG_MH_INTERP_WAVES", raising=False)
        eng = E.Engine(cp, C, seed=17, chain_offset=4)
        d = eng.device_alloc(max(1, ns * cp.S * C) * 8)
        st = eng.mh_run(ns, nw, ov, rec, d)
        draws = eng.download(d, (ns, cp.S, C), dtype=np.int64)
        eng.device_free(d)
        out.append((draws, eng.get_values(), eng.mh_scales(), eng.mh_log_weight(), st.accept_rate))
        eng.close()
    assert 0.0 < out[0][4] < 1.0
    for o in out[1:]:
        for a, b in zip(out[0], o):
            assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
