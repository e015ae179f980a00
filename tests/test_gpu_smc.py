"""GPU parity: likelihood-tempered SMC (adaptive_smc, src/inference/smc.rs) and its
population-wide primitives (log-sum-exp, next_beta, the three resamplers) through the C ABI
against the CPU oracle.  Resample indices are integer work: exact."""
import math

import numpy as np
from tests import knife
import pytest

from fugue_amd import engine as E
from fugue_amd import workloads as W

pytestmark = pytest.mark.gpu


def test_device_log_sum_exp(oracle):
    rng = np.random.default_rng(0)
    for n in (1, 7, 513, 100_003):
        x = rng.normal(-50, 30, size=n)
        assert E.device_log_sum_exp(x) == pytest.approx(oracle.log_sum_exp(x), rel=1e-13, abs=1e-13)
    # reference known answers (src/core/numerical.rs:152-203, tests/f_dist_numerical.rs:22-85)
    assert E.device_log_sum_exp([700.0, 701.0, 699.0]) == pytest.approx(701.4076059644444, abs=1e-12)
    assert E.device_log_sum_exp([1000.0, 1001.0]) == pytest.approx(1001.3132616875182, abs=1e-12)
    assert E.device_log_sum_exp([-math.inf] * 3) == -math.inf
    assert E.device_log_sum_exp([-math.inf, 0.0]) == 0.0
    assert E.device_log_sum_exp([]) == -math.inf


def test_device_next_beta(oracle):
    rng = np.random.default_rng(1)
    for n, beta, scale in ((1000, 0.0, 5.0), (40_000, 0.3, 2.0), (40_000, 0.0, 0.01)):
        ll = rng.normal(-3, scale, size=n)
        lw = np.full(n, -math.log(n))
        tgt = 0.5 * n
        assert E.device_next_beta(beta, lw, ll, tgt) == pytest.approx(oracle.next_beta(beta, lw, ll, tgt), rel=1e-10)
    assert E.device_next_beta(0.0, lw, ll, tgt) == 1.0          # flat likelihood: jump straight to 1 (smc.rs:604-607)


@pytest.mark.parametrize("method", [E.RESAMPLE_SYSTEMATIC, E.RESAMPLE_STRATIFIED, E.RESAMPLE_MULTINOMIAL])
@pytest.mark.parametrize("n", [5, 1000, 6001])
def test_resample_indices_exact(oracle, method, n):
    rng = np.random.default_rng(n + method)
    w = rng.gamma(0.5, size=n)
    w /= w.sum()
    u = rng.random(n)
    if method == E.RESAMPLE_SYSTEMATIC:
        got, exp = E.device_resample_indices(method, w, u[:1]), oracle.systematic_indices(w, u[0])
    elif method == E.RESAMPLE_STRATIFIED:
        got, exp = E.device_resample_indices(method, w, u), oracle.stratified_indices(w, u)
    else:
        got, exp = E.device_resample_indices(method, w, u), oracle.multinomial_indices(w, u)
    # the parallel prefix sum associates the additions differently from the reference's sequential
    # walk: an index may move by one only where a threshold sits within ~1e-13 of a cumulative weight
    diff = np.nonzero(got != exp)[0]
    cum = np.cumsum(w)
    for j in diff:
        thr = (u[0] / n + j / n) if method == 1 else ((j + u[j]) / n if method == 2 else u[j])
        assert abs(int(got[j]) - int(exp[j])) == 1 and abs(cum[min(got[j], exp[j])] - thr) < 1e-12
        knife.used("resampling indices", method=method, slot=int(j), gpu=int(got[j]), oracle=int(exp[j]), threshold_minus_cum=float(thr - cum[min(got[j], exp[j])]))
    assert len(diff) <= 1


def test_resample_large_population_properties():
    """BASELINE C4 size: 1 048 576 particles.  Systematic resampling keeps every particle
    floor(N w) or ceil(N w) times and the indices are sorted."""
    n = 1 << 20
    rng = np.random.default_rng(4)
    w = rng.gamma(2.0, size=n)
    w /= w.sum()
    idx = E.device_resample_indices(E.RESAMPLE_SYSTEMATIC, w, [0.37])
    assert (np.diff(idx) >= 0).all() and idx.min() >= 0 and idx.max() < n
    counts = np.bincount(idx, minlength=n)
    assert np.abs(counts - n * w).max() < 1.0 + 1e-6


def _smc_pair(oracle, prog, n, seed, **kw):
    cp, om = E.compile_model(prog), oracle.OracleModel(prog)
    eng = E.Engine(cp, n, seed=seed)
    seq = kw.get("sequential", False)
    got = eng.smc_run(rejuvenation_steps=kw.get("R", 0), ess_threshold=kw.get("thr", 0.5), resampling_method=kw.get("method", 1), sequential_adaptation=seq)
    exp = om.smc_run(n, seed, method=kw.get("method", 1), ess_threshold=kw.get("thr", 0.5), rejuvenation_steps=kw.get("R", 0), batched=0 if seq else 1)
    return cp, got, exp


def test_smc_importance_sampling_matches_oracle(oracle):
    """rejuvenation_steps == 0: a single importance-sampling reweight (smc.rs:484-493)."""
    cp, got, exp = _smc_pair(oracle, W.smc_normal(), 4096, seed=42)
    assert got["log_evidence"] == pytest.approx(exp["log_evidence"], rel=1e-12)
    np.testing.assert_allclose(got["values"].view(np.float64), exp["values"].view(np.float64), rtol=1e-11)
    np.testing.assert_allclose(got["weights"], exp["weights"], rtol=1e-9, atol=1e-300)
    assert list(got["betas"]) == [1.0]


@pytest.mark.parametrize("method", [E.RESAMPLE_SYSTEMATIC, E.RESAMPLE_STRATIFIED, E.RESAMPLE_MULTINOMIAL])
def test_smc_tempered_matches_oracle(oracle, method):
    """The full ladder (next_beta bisection, reweight, evidence, resample, gather, tempered
    rejuvenation with the per-sweep adaptation) against the oracle in its batched-adaptation mode."""
    cp, got, exp = _smc_pair(oracle, W.smc_normal(), 3000, seed=42, R=3, method=method)
    np.testing.assert_allclose(got["betas"], exp["betas"], rtol=1e-9)
    assert got["log_evidence"] == pytest.approx(exp["log_evidence"], rel=1e-9)
    g, o = got["values"].view(np.float64)[0], exp["values"].view(np.float64)[0]
    bad = ~np.isclose(g, o, rtol=1e-9, atol=1e-12)
    if bad.any(): knife.used("adaptive_smc: particle values", method=method, particles=np.nonzero(bad)[0].tolist())
    assert bad.sum() <= 3, bad.sum()                   # a knife-edge accept / resample boundary
    np.testing.assert_allclose(got["weights"][~bad], exp["weights"][~bad], rtol=1e-8)
    assert got["n_model_runs"] == exp["n_model_evals"]


@pytest.mark.parametrize("name", ["smc_normal", "refmodel8"])
def test_smc_sequential_adaptation_is_the_references_order(oracle, name):
    """fg_smc_config.sequential_adaptation: particle-major moves with ONE DiminishingAdaptation updated after every move
    (smc.rs:482,544-553,698-713) -- the reference's own semantics, against the oracle's unbatched form: the beta ladder, the
    evidence, every particle and weight.  (The default, batched per sweep, is the documented deviation; this mode closes it.)"""
    from tests.models import ZOO
    prog = W.smc_normal() if name == "smc_normal" else ZOO[name]()
    cp, got, exp = _smc_pair(oracle, prog, 1500, seed=42, R=3, sequential=True)
    np.testing.assert_allclose(got["betas"], exp["betas"], rtol=1e-9)
    assert got["log_evidence"] == pytest.approx(exp["log_evidence"], rel=1e-9)
    bad = np.zeros(1500, dtype=bool)
    for j in range(cp.S):
        bad |= ~np.isclose(got["values"].view(np.float64)[j], exp["values"].view(np.float64)[j], rtol=1e-9, atol=1e-12)
    # an accept on a knife edge changes that particle AND, through the shared adaptation, every later scale by ~1e-4: the comparison is
    # exact or it visibly is not
    assert bad.sum() == 0, bad.sum()
    np.testing.assert_allclose(got["weights"], exp["weights"], rtol=1e-8)
    assert got["n_model_runs"] == exp["n_model_evals"]
    _, got_b, _ = _smc_pair(oracle, prog, 1500, seed=42, R=3)
    assert not np.array_equal(got_b["values"], got["values"])            # ... and it is not the batched form


@pytest.mark.parametrize("n_sites", [6, 40])
def test_smc_multisite_model_matches_oracle(oracle, n_sites):
    """(40 sites: sixteen tiles of 41 rows do not fit a block's LDS -- the rejuvenation kernel takes the tiles per block that do;
    round 3 returned FG_E_LIMIT there)"""
    cp, got, exp = _smc_pair(oracle, W.reference_model(n_sites), 2048, seed=7, R=2)
    np.testing.assert_allclose(got["betas"], exp["betas"], rtol=1e-8)
    assert got["log_evidence"] == pytest.approx(exp["log_evidence"], rel=1e-8)
    g, o = got["values"].view(np.float64), exp["values"].view(np.float64)
    bad = (~np.isclose(g, o, rtol=1e-8, atol=1e-11)).any(axis=0)
    assert bad.sum() <= 4, bad.sum()


def test_smc_million_particles_closed_form():
    """BASELINE C4: mu ~ N(0,1), y ~ N(mu, 0.5) = 1.5 with 1 048 576 particles, Systematic / 0.5 /
    3 rejuvenation moves (examples/smc_inference.rs:36-65): posterior N(1.2, 0.2),
    log Z = -1.9305103088617774."""
    cp = E.compile_model(W.smc_normal())
    n = 1 << 20
    eng = E.Engine(cp, n, seed=42)
    r = eng.smc_run(rejuvenation_steps=3)
    mu = r["values"].view(np.float64)[0]
    mean = float((r["weights"] * mu).sum())
    var = float((r["weights"] * (mu - mean) ** 2).sum())
    assert abs(r["log_evidence"] - (-1.9305103088617774)) < 5e-3
    assert abs(mean - 1.2) < 5e-3 and abs(var - 0.2) < 5e-3
    assert abs(r["weights"].sum() - 1.0) < 1e-9 and len(r["betas"]) >= 2 and r["betas"][-1] == 1.0


def test_standalone_smc_building_blocks(oracle):
    """smc_prior_particles / normalize_particles / effective_sample_size / resample_particles / rejuvenate_particles
    (smc.rs:230-233, 326-349, 698-790) as separate calls over the engine's population."""
    N = 5000
    cp = E.compile_model(W.smc_normal())
    eng = E.Engine(cp, N, seed=42)
    with pytest.raises(E.EngineError):
        eng.smc_ess()                                               # no population yet
    eng.smc_prior_particles()
    lw, w = eng.smc_weights()
    mu = eng.get_values()[0].view(np.float64)
    ll = -0.5 * ((1.5 - mu) / 0.5) ** 2 - np.log(0.5) - 0.5 * np.log(2 * np.pi)
    assert np.allclose(lw, ll, rtol=1e-12)                          # log_weight = log-likelihood only (FG-03)
    ref = np.exp(lw - oracle.log_sum_exp(lw)); ref /= ref.sum()
    assert np.allclose(w, ref, rtol=1e-12) and abs(w.sum() - 1.0) < 1e-12
    assert eng.smc_ess() == pytest.approx(1.0 / np.sum(w * w), rel=1e-12)
    assert eng.smc_ess() == pytest.approx(oracle.ess_particles(w), rel=1e-12)
    assert abs(np.sum(w * mu) - 1.2) < 0.05                         # importance-sampling posterior mean of C4
    # all log-weights -inf -> uniform fallback (smc.rs:735-741)
    eng.smc_set_log_weights(np.full(N, -np.inf)); eng.smc_normalize()
    assert np.array_equal(eng.smc_weights()[1], np.full(N, 1.0 / N))
    eng.smc_set_log_weights(lw); eng.smc_normalize()
    assert np.allclose(eng.smc_weights()[1], ref, rtol=1e-12)
    # resample: the ancestors are those of the tested index primitive for the same uniform; clones; weights exactly 1/N
    before = eng.get_values().copy()
    idx = eng.smc_resample(E.RESAMPLE_SYSTEMATIC, step=3)
    import ctypes
    st = oracle.stream(42, 0, 3, 5)                                 # (seed, chain 0, step, FG_RNG_SMC_RESAMPLE)
    oracle.lib().orc_stream_u01.restype = ctypes.c_double
    U = oracle.lib().orc_stream_u01(ctypes.byref(st))
    assert np.array_equal(idx, E.device_resample_indices(E.RESAMPLE_SYSTEMATIC, w, U))
    assert idx.min() >= 0 and idx.max() < N and (np.diff(idx) >= 0).all()
    assert np.array_equal(eng.get_values(), before[:, idx])
    lw2, w2 = eng.smc_weights()
    assert np.array_equal(w2, np.full(N, 1.0 / N)) and np.array_equal(lw2, np.full(N, np.log(1.0 / N)))
    assert eng.smc_ess() == pytest.approx(N, rel=1e-9)
    # rejuvenation moves particles, leaves the weights exactly uniform (tests/f_smc_smc.rs:45-205) and targets pi_1
    acc = eng.smc_rejuvenate(1.0, 5)
    assert 0.1 < acc < 0.95
    assert np.array_equal(eng.smc_weights()[1], np.full(N, 1.0 / N))
    mu2 = eng.get_values()[0].view(np.float64)
    assert (mu2 != before[0, idx].view(np.float64)).mean() > 0.3
    assert abs(mu2.mean() - 1.2) < 0.05 and abs(mu2.var() - 0.2) < 0.05


def test_next_beta_zoom_passes_agree_with_the_plain_passes():
    """next_beta brackets the ESS crossing by zoom passes and replays the reference's 64 halvings from the bracket;
    FG_SMC_ZOOM=0 evaluates every halving (three levels per pass).  Both follow the same comparisons `ESS(mid) < target` wherever the
    evaluated ESS is monotone: the ladders agree to a few units in the last place, the evidence to rounding.  (The switch is read once per
    process: tools/ab_smc_zoom.py runs each setting in a child process -- six models, ladders of 2 to ~20 steps, up to 1 048 576 particles.)"""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "ab_smc_zoom.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rows = re.findall(r"^(\S+)\s+steps (\d+)/(\d+)\s+max \|d beta\| (\S+)\s+d logZ (\S+)", out.stdout, flags=re.M)
    assert len(rows) == 6, out.stdout
    for name, s0, s1, db, dz in rows:
        assert s0 == s1, (name, s0, s1)
        assert float(db) <= 1e-14 and float(dz) <= 1e-12, (name, db, dz)


def test_smc_separate_kernels_path_agrees_with_the_fused_launch(monkeypatch):
    """What follows next_beta's last pass is one launch (k_smc_ess2_apply) unless beta' = beta + 1e-9 -- no candidate of any pass -- when the
    separate maximum / sum / finish / apply / scan kernels take the step.  FG_SMC_FORCE_SUM=1 sends every step down that path: the same
    ladder, the evidence and the weights to rounding (the fused launch takes the reweight's sum from the pass that evaluated beta')."""
    cp = E.compile_model(W.normal_sites(8))
    out = []
    for force in ("0", "1"):
        monkeypatch.setenv("FG_SMC_FORCE_SUM", force)
        eng = E.Engine(cp, 20000, seed=9)
        r = eng.smc_run(rejuvenation_steps=2, ess_threshold=0.5)
        out.append(r)
        eng.close()
    a, b = out
    assert len(a["betas"]) >= 3 and np.array_equal(a["betas"], b["betas"])
    assert a["log_evidence"] == pytest.approx(b["log_evidence"], rel=1e-13)
    bad = ~np.isclose(a["values"].view(np.float64), b["values"].view(np.float64), rtol=1e-12, atol=0).all(axis=0)
    assert bad.sum() <= 2, int(bad.sum())                 # (a resampling threshold within rounding of a cumulative weight)
    np.testing.assert_allclose(a["weights"][~bad], b["weights"][~bad], rtol=1e-10)


@pytest.mark.parametrize("method", [0, 1, 2])
def test_adaptive_smc_edge_population_sizes(method):
    """Populations around every grain of the kernels -- one particle, a wave, a scan chunk of 2 048, their neighbours, a prime -- for the
    three resamplers: a finite evidence, weights that sum to one, a strictly increasing ladder that ends at one."""
    cp = E.compile_model(W.normal_sites(8))
    for N in (1, 2, 63, 64, 65, 2047, 2048, 2049, 4097, 100003):
        eng = E.Engine(cp, N, seed=11)
        r = eng.smc_run(rejuvenation_steps=2, ess_threshold=0.5, resampling_method=method)
        eng.close()
        assert np.isfinite(r["log_evidence"]), N
        assert abs(r["weights"].sum() - 1.0) < 1e-9 and (r["weights"] >= 0).all(), N
        assert r["betas"][-1] == 1.0 and (np.diff(r["betas"]) > 0).all(), (N, r["betas"])
