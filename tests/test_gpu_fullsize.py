"""BASELINE.json's configurations at their FULL sizes, checked through properties that do not need the
oracle to run at that size: closed-form posteriors, independence of results from how chains are sharded
over engines (the multi-GPU layout), agreement with a numpy evaluation of the same density, and the
oracle on a handful of chains of the full-size model."""
import numpy as np
from tests import knife
import pytest

from fugue_amd import diagnostics as D
from fugue_amd import engine as E
from fugue_amd import workloads as W

pytestmark = pytest.mark.gpu


def _hmc_draws(cp, C, ns, nw, seed, chain_offset=0, **cfg):
    eng = E.Engine(cp, C, seed=seed, chain_offset=chain_offset)
    d = eng.device_alloc(ns * cp.d * C * 8)
    st = eng.hmc_run(E.hmc_config(**cfg), ns, nw, d)
    prov = D.EngineMoments(eng, d, ns, cp.d)
    cd = D.ChainDiagnostics(prov)
    rhat, mean = cd.split_rhat(), cd.pooled_mean()
    prov.close()
    return eng, d, st, rhat, mean


def test_c2_readme_model_65536_chains_1000_steps():
    """configs[1]: conjugate Normal-Normal (README mu model), hmc_chain, 65 536 chains x 1 000 steps.
    Posterior N(0.96, 0.2); all 65 536 chains enter split-R-hat."""
    cp = E.compile_model(W.readme_normal())
    C, ns = 65536, 1000
    eng, d, st, rhat, mean = _hmc_draws(cp, C, ns, 300, seed=7, grad_mode=E.GRAD_FD_DENSE)
    assert abs(mean[0] - 0.96) < 1e-3                               # north_star: within 1e-3 of the closed form
    assert abs(rhat[0] - 1.0) < 0.01
    assert 0.6 < st.accept_rate <= 1.0 and st.n_divergent == 0
    last = eng.download(d + (ns - 1) * C * 8, (C,))
    assert abs(last.var() - 0.2) < 0.01                             # a single draw across the chains is a posterior sample
    eng.device_free(d)


def test_target_model_65536_chains_and_sharding_invariance():
    """north_star target model (32-site Normal) at 65 536 chains: posterior mean within 1e-3, and the draws of
    chains [32768, 65536) do not depend on whether they ran in one engine of 65 536 chains or in their own engine
    with chain_offset = 32768 -- the property that makes `--gpus N` a pure re-partition (RNG keyed by global chain id)."""
    cp = E.compile_model(W.normal_sites(32))
    C, ns, nw = 65536, 60, 120
    eng, d, st, rhat, mean = _hmc_draws(cp, C, ns, nw, seed=1, grad_mode=E.GRAD_FD_SPARSE)
    _, tm, _ = W.normal_sites_truth(32)
    assert np.abs(mean - tm).max() < 1e-3
    assert np.abs(rhat - 1.0).max() < 0.02 and st.n_divergent == 0      # halves of 30 draws: (n-1)/n alone is 0.967
    whole = eng.download(d, (ns, cp.d, C))[:, :, C // 2:]
    eng.device_free(d)
    eng2, d2, *_ = _hmc_draws(cp, C // 2, ns, nw, seed=1, chain_offset=C // 2, grad_mode=E.GRAD_FD_SPARSE)
    part = eng2.download(d2, (ns, cp.d, C // 2))
    eng2.device_free(d2)
    assert np.array_equal(whole, part)


def test_c3_regression_1024_observations_32_coefficients(oracle):
    """configs[2] model at full size (32 Normal coefficients, 1 024 observations; the FG_OP_DOT path): the log-joint
    against a numpy evaluation of the same density, and one HMC transition per chain against the oracle."""
    X, y, _ = W.ridge_data(1024, 32)
    prog = W.ridge_regression(X, y)
    cp, om = E.compile_model(prog), oracle.OracleModel(prog)
    assert cp.n_instructions < 5000                                  # 4 instructions per observation, not 35
    C = 64
    rng = np.random.default_rng(2)
    beta = 0.3 * rng.standard_normal((cp.d, C))
    order = [int(n.split("#")[1]) for n in cp.site_names]            # site j holds beta#order[j] (lexicographic order)
    eng = E.Engine(cp, C, seed=4)
    cells = np.ascontiguousarray(beta).view(np.int64)
    eng.set_values(cells)
    acc = eng.log_joint()
    b_model = np.empty_like(beta)
    b_model[order] = beta                                           # beta#j for the design matrix columns
    resid = (y[:, None] - X @ b_model) / 0.5
    lik = (-0.5 * resid ** 2 - np.log(0.5) - 0.5 * np.log(2 * np.pi)).sum(axis=0)
    pri = (-0.5 * beta ** 2 - 0.5 * np.log(2 * np.pi)).sum(axis=0)
    assert np.allclose(acc[0], pri, rtol=1e-12) and np.allclose(acc[1], lik, rtol=1e-10)
    for c in range(0, C, 9):
        oacc, _ = om.run_score(cells[:, c])
        assert np.allclose(acc[:, c], oacc, rtol=1e-12, atol=1e-12)
    # one transition (L = 3) from these positions with injected momentum and uniform: same decision, same position
    p0 = rng.standard_normal((cp.d, C))
    u = rng.random(C)
    accd, alpha, div, lj = eng.hmc_transition_injected(E.hmc_config(grad_mode=E.GRAD_FD_SPARSE, n_leapfrog=3), 0.002, p0, u)
    newv = eng.get_values().view(np.float64)
    for c in range(0, C, 16):
        q = beta[:, c].copy()
        qo, ljo, oacc, oalpha, odiv = om.hmc_transition(cells[:, c], q, om.log_joint_at(cells[:, c], q), 0.002, 3, p0[:, c].copy(), float(u[c]))
        assert bool(accd[c]) == oacc and bool(div[c]) == odiv
        assert abs(alpha[c] - oalpha) < 1e-5
        assert np.allclose(newv[:, c], qo, rtol=1e-6, atol=1e-8)


def test_c3_regression_65536_chains_closed_form_ridge_posterior():
    """configs[2] at its full chain count: 32 coefficients x 1 024 observations, hmc_chain on 65 536 chains (the per-node
    total of the 8 x 8 192 sharding).  Pooled posterior mean within 1e-3 of the closed-form ridge posterior
    mu = Sigma X'y / sigma^2, Sigma = (X'X / sigma^2 + lambda I)^-1, pooled variance against
    diag(Sigma), split-R-hat over all 65 536 chains < 1.01."""
    X, y, _ = W.ridge_data(1024, 32)
    cp = E.compile_model(W.ridge_regression(X, y))
    assert cp.stream_records[0] == 32 * 1025 and cp.stream_records[2] == 1      # linear-predictor records: the multi-wave stream kernel
    C, nw, ns = 65536, 110, 40
    eng = E.Engine(cp, C, seed=5)
    d = eng.device_alloc(ns * cp.d * C * 8)
    st = eng.hmc_run(E.hmc_config(grad_mode=E.GRAD_FD_SPARSE), ns, nw, d)
    prov = D.EngineMoments(eng, d, ns, cp.d)
    cd = D.ChainDiagnostics(prov)
    rhat, mean = cd.split_rhat(), cd.pooled_mean()
    prov.close()
    mu, Sig = W.ridge_truth(X, y)
    order = [cp.site_names.index(f"beta#{j}") for j in range(32)]
    assert np.abs(mean[order] - mu).max() < 1e-3, np.abs(mean[order] - mu).max()
    assert rhat.max() < 1.01 and rhat.min() > 0.97, (rhat.min(), rhat.max())   # (n-1)/n of 20-draw halves alone is 0.975
    last = eng.download(d + (ns - 1) * cp.d * C * 8, (cp.d, C))                # one draw across the chains = 65 536 posterior samples
    assert np.allclose(last.var(axis=1)[order], np.diag(Sig), rtol=0.05)
    assert st.n_divergent == 0 and 0.6 < st.accept_rate <= 1.0
    eng.device_free(d)


def test_c5_mixture_262144_chains():
    """configs[4] as SURVEY 8d defines it: 4-component Gaussian mixture, N_data = 64 (S = 68: 4 f64 + 64 usize sites,
    O = 64), adaptive_mcmc_chain at 262 144 chains.  Assignments stay in {0..3}, the sorted means recover the generating
    (-6,-2,2,6), and two engines of 131 072 chains reproduce the second half exactly (the multi-GPU re-partition)."""
    data, _ = W.mixture_data(64)
    cp = E.compile_model(W.mixture(data))
    assert cp.S == 68 and cp.d == 4 and cp.O == 64
    C, nw, ns = 262144, 2000, 10
    mu_sites = [cp.site_names.index(f"mu#{k}") for k in range(4)]
    z_sites = [j for j, n in enumerate(cp.site_names) if n.startswith("z#")]
    rec = mu_sites + z_sites
    eng = E.Engine(cp, C, seed=11)
    d = eng.device_alloc(ns * len(rec) * C * 8)
    st = eng.mh_run(ns, nw, None, rec, d)
    cells = eng.download(d, (ns, len(rec), C), dtype=np.int64)
    eng.device_free(d)
    z = cells[:, 4:, :]
    assert z.min() >= 0 and z.max() <= 3                            # Categorical indices are exact integers
    mu = np.sort(cells[:, :4, :].view(np.float64), axis=1)
    med = np.median(mu.mean(axis=0), axis=1)
    assert np.abs(med - np.array([-6.0, -2.0, 2.0, 6.0])).max() < 0.6, med
    assert 0.05 < st.accept_rate < 0.9
    eng2 = E.Engine(cp, C // 2, seed=11, chain_offset=C // 2)
    d2 = eng2.device_alloc(ns * len(rec) * (C // 2) * 8)
    eng2.mh_run(ns, nw, None, rec, d2)
    half = eng2.download(d2, (ns, len(rec), C // 2), dtype=np.int64)
    eng2.device_free(d2)
    assert np.array_equal(cells[:, :, C // 2:], half)


def test_c5_responsibilities_at_the_true_means_262144_chains():
    """SURVEY 8d ground truth for C5: with the component means held at their generating values the posterior of every
    assignment z#i is the exact responsibility vector r_ik = N(x_i; mu_k, 1) / sum_k N(x_i; mu_k, 1).  262 144 independent
    chains of adaptive_mcmc_chain (prior-resample proposals on the usize sites, mh.rs:516-530) give 262 144 independent
    draws of each z#i: the observed frequencies match r_ik within 5 binomial standard errors."""
    from fugue_amd import model as M
    data, _ = W.mixture_data(64)
    means = np.array([-6.0, -2.0, 2.0, 6.0])
    P = M.Program()
    for i, xi in enumerate(data):
        z = P.sample(M.addr("z", i), M.Categorical([0.25] * 4))
        P.observe(M.addr("x", i), M.Normal(M.select(z, [float(m) for m in means]), 1.0), float(xi))
    cp = E.compile_model(P)
    assert cp.stream_records[1] == 128                               # Categorical-table and option-select records: the score stream
    C, nw = 262144, 64 * 60
    rec = list(range(cp.S))
    eng = E.Engine(cp, C, seed=23)
    d = eng.device_alloc(len(rec) * C * 8)
    eng.mh_run(1, nw, None, rec, d)
    z = eng.download(d, (1, cp.S, C), dtype=np.int64)[0]
    eng.device_free(d)
    assert z.min() >= 0 and z.max() <= 3
    logr = -0.5 * (data[:, None] - means[None, :]) ** 2
    r = np.exp(logr - logr.max(axis=1, keepdims=True))
    r /= r.sum(axis=1, keepdims=True)
    for j, name in enumerate(cp.site_names):
        i = int(name.split("#")[1])
        freq = np.bincount(z[j], minlength=4) / C
        se = np.sqrt(np.maximum(r[i] * (1.0 - r[i]), 1e-12) / C)
        assert (np.abs(freq - r[i]) <= 5.0 * se + 2e-5).all(), (name, freq, r[i])


def test_c5_mixture_64_points_categorical_indices_exact_vs_oracle(oracle):
    """The full-size C5 model (S = 68) on 64 chains against the CPU oracle on the same Philox streams: every recorded
    Categorical index identical, every mean to 1e-9."""
    data, _ = W.mixture_data(64)
    prog = W.mixture(data)
    cp, om = E.compile_model(prog), oracle.OracleModel(prog)
    C, nw, ns = 64, 400, 40
    rec = list(range(cp.S))
    eng = E.Engine(cp, C, seed=3, chain_offset=5)
    d = eng.device_alloc(ns * cp.S * C * 8)
    eng.mh_run(ns, nw, None, rec, d)
    draws = eng.download(d, (ns, cp.S, C), dtype=np.int64)
    eng.device_free(d)
    odraws, ofinal, _, _ = om.mh_run(3, C, nw, ns, None, rec, chain0=5, n_threads=8)
    bad = np.zeros(C, dtype=bool)
    for j in range(cp.S):
        if cp.site_vtypes[j] == 0:
            bad |= (~np.isclose(draws[:, j].view(np.float64), odraws[:, j].view(np.float64), rtol=1e-9, atol=1e-12)).any(axis=0)
        else:
            bad |= (draws[:, j] != odraws[:, j]).any(axis=0)
    if bad.any(): knife.used("C5 mixture at full size: MH draws", chains=np.nonzero(bad)[0].tolist())
    assert bad.sum() <= 1, np.nonzero(bad)[0]                       # <= 1 acceptance on a 1e-13 knife edge


def test_models_larger_than_one_lds_tile(oracle):
    """hmc_chain has no size limit in the reference (hmc.rs:238-260).  A program whose 64-chain tile exceeds the 160 KB of a CU
    keeps its tile in a global scratch and runs on the one-wave-per-tile kernels: 400 independent sites (tile 419 KB) and a
    200-coefficient regression (215 KB) run the log-joint, HMC transitions under injected momentum / uniform against the oracle,
    a short adaptive hmc_run, single-site MH, a recorded trajectory (step_recorded) and adaptive_smc with rejuvenation."""
    rng = np.random.default_rng(2)
    X, y, _ = W.ridge_data(30, 200)
    for name, prog in (("normal_sites(400)", W.normal_sites(400)), ("ridge 200 x 30", W.ridge_regression(X, y))):
        cp, om = E.compile_model(prog), oracle.OracleModel(prog)
        C = 96
        eng = E.Engine(cp, C, seed=2)
        eng.prior_init()
        cells = eng.get_values()
        acc = eng.log_joint()
        for c in (0, 17, 95):
            oacc, _ = om.run_score(cells[:, c])
            assert np.allclose(acc[:, c], oacc, rtol=1e-12, atol=1e-12), name
        p0, u = rng.standard_normal((cp.d, C)), rng.random(C)
        cfg = E.hmc_config(n_leapfrog=3)
        eps = 0.02
        acc_, alpha, div, lj = eng.hmc_transition_injected(cfg, eps, p0, u)
        nxt = eng.get_values()
        for c in (0, 40, 95):
            q = np.ascontiguousarray(cells[om.f64_sites, c]).view(np.float64)
            qo, ljo, oacc, oalpha, odiv = om.hmc_transition(cells[:, c], q, om.log_joint_at(cells[:, c], q), eps, 3, p0[:, c], u[c])
            assert odiv == bool(div[c]), name
            if not odiv:
                assert abs(alpha[c] - oalpha) < 1e-5 and (oacc == bool(acc_[c]) or abs(u[c] - oalpha) < 1e-5), name
                if oacc == bool(acc_[c]):     # 1e-6: the reference's own FD noise (1 ulp of a log-joint of several hundred / 2h) through three kicks
                    assert np.allclose(np.ascontiguousarray(nxt[om.f64_sites, c]).view(np.float64), qo, rtol=1e-6, atol=1e-6), name
        d = eng.device_alloc(6 * cp.d * C * 8)
        st = eng.hmc_run(E.hmc_config(n_leapfrog=4), 6, 12, d)
        assert "k_hmc_steps" in eng.hmc_last_kernel(), eng.hmc_last_kernel()
        assert np.isfinite(eng.download(d, (6, cp.d, C))).all() and st.n_divergent == 0 and st.accept_rate > 0.3, name
        eng.device_free(d)
        st = eng.mh_run(40, 40, None, [0, cp.S - 1], buf := eng.device_alloc(40 * 2 * C * 8))
        assert 0.05 < st.accept_rate < 0.95, name
        eng.device_free(buf)
        eng.close()
        # step_recorded (hmc.rs:811-817) from a tile in global memory: RNG-neutral, and every point the oracle's leapfrog
        rec, plain = E.Engine(cp, C, seed=3), E.Engine(cp, C, seed=3)
        cfg = E.hmc_config(n_leapfrog=3, init_step_size=0.01)
        rec.hmc_init(cfg, 0); plain.hmc_init(cfg, 0)
        ids = [0, 63, 64, 95]
        before = rec.get_values()
        traj, ham, npts = rec.hmc_step_recorded(ids, 3)
        plain.hmc_step(1)
        assert np.array_equal(rec.get_values(), plain.get_values()), name
        for k, c in enumerate(ids):
            q0 = np.ascontiguousarray(before[om.f64_sites, c]).view(np.float64)
            assert np.array_equal(traj[k, 0], q0) and npts[k] == 4 and np.isfinite(ham[k]).all(), name
            p0k, _ = oracle.hmc_momentum(3, c, 0, cp.d)
            qo, po, dv = om.leapfrog(before[:, c], q0.copy(), p0k.copy(), 0.01, 3)
            assert not dv and np.allclose(traj[k, 3], qo, rtol=1e-6, atol=1e-6), name
        rec.close(); plain.close()
        # adaptive_smc has no size limit either (smc.rs:455-581, 631-713): the beta ladder, the evidence and the particles of a run with
        # rejuvenation against the oracle (batched adaptation on both sides), tile and block histogram in global memory
        n = 512
        eng = E.Engine(cp, n, seed=11)
        got = eng.smc_run(rejuvenation_steps=2, ess_threshold=0.5, resampling_method=E.RESAMPLE_SYSTEMATIC)
        exp = om.smc_run(n, 11, method=E.RESAMPLE_SYSTEMATIC, ess_threshold=0.5, rejuvenation_steps=2, batched=1)
        assert len(got["betas"]) == len(exp["betas"]) > 3, (name, len(got["betas"]), len(exp["betas"]))
        np.testing.assert_allclose(got["betas"], exp["betas"], rtol=1e-9, err_msg=name)
        assert got["log_evidence"] == pytest.approx(exp["log_evidence"], rel=1e-9), name
        g, o = got["values"].view(np.float64), exp["values"].view(np.float64)
        bad = (~np.isclose(g, o, rtol=1e-8, atol=1e-11)).any(axis=0)
        if bad.any(): knife.used("adaptive_smc beyond one LDS tile: particle values", model=name, particles=np.nonzero(bad)[0][:8].tolist())
        assert bad.sum() <= 4, (name, int(bad.sum()))                 # a knife-edge accept / resample boundary
        assert got["n_model_runs"] == exp["n_model_evals"], name
        eng.close()
        # ... and in the reference's own sequential order (one wave walks the particles, its tile in global memory): a small population
        n = 64
        eng = E.Engine(cp, n, seed=12)
        got = eng.smc_run(rejuvenation_steps=1, ess_threshold=0.5, resampling_method=E.RESAMPLE_SYSTEMATIC, sequential_adaptation=True)
        exp = om.smc_run(n, 12, method=E.RESAMPLE_SYSTEMATIC, ess_threshold=0.5, rejuvenation_steps=1, batched=0)
        assert len(exp["betas"]) > 2, name
        np.testing.assert_allclose(got["betas"], exp["betas"], rtol=1e-9, err_msg=name)
        assert got["log_evidence"] == pytest.approx(exp["log_evidence"], rel=1e-9), name
        eng.close()


@pytest.mark.parametrize("name", ["normal32", "ridge", "hier_scale", "alldists", "mixture"])
def test_global_tile_instantiations_change_no_result(name, monkeypatch):
    """The global-tile instantiations of the one-wave kernels (prior, log-joint, HMC, MH) against their LDS instantiations on
    programs that fit both: the same cells, draws, step sizes and MH chains, bit for bit."""
    from tests.models import ZOO
    cp = E.compile_model(ZOO[name]())
    C = 100
    out = []
    for gt in ("0", "1"):
        monkeypatch.setenv("FG_GLOBAL_TILE", gt)
        monkeypatch.setenv("FG_HMC_SEP", "0"); monkeypatch.setenv("FG_HMC_LIN", "0"); monkeypatch.setenv("FG_HMC_WAVES", "1"); monkeypatch.setenv("FG_MH_MW", "0")
        eng = E.Engine(cp, C, seed=4, chain_offset=3)
        eng.prior_init()
        cells, acc = eng.get_values(), eng.log_joint()
        res = [cells, acc]
        if cp.d > 0:
            d = eng.device_alloc(8 * cp.d * C * 8)
            st = eng.hmc_run(E.hmc_config(n_leapfrog=5), 8, 10, d)
            res += [eng.download(d, (8, cp.d, C)), eng.hmc_step_sizes(), eng.hmc_log_joint(), st.accept_rate]
            eng.device_free(d)
        buf = eng.device_alloc(30 * cp.S * C * 8)
        st = eng.mh_run(30, 30, None, list(range(cp.S)), buf)
        res += [eng.download(buf, (30, cp.S, C), dtype=np.int64), eng.mh_scales(), st.accept_rate]
        eng.device_free(buf)
        eng.close()
        out.append(res)
    for a, b in zip(*out):
        assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
