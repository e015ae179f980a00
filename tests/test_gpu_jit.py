"""Models compiled at run time (fg_jit.cpp: one C++ statement per interpreter instruction, hiprtc, the multi-wave HMC kernel around
the generated functions) against the interpreter kernels: the same operations in the same order, so draws, step sizes, mass
matrix, log-joint, values and statistics agree BIT FOR BIT -- and against the oracle like every other path."""
import numpy as np
import pytest

from fugue_amd import engine as E
from tests.models import ZOO

pytestmark = pytest.mark.gpu

INTERP_MODELS = ["alldists", "poisson_glm", "hier_logsigma", "logistic", "coin"]


@pytest.mark.parametrize("name,adapt_mass", [("alldists", True), ("alldists", False), ("poisson_glm", True), ("hier_logsigma", False), ("logistic", True), ("coin", False)])
def test_jit_hmc_is_bit_identical_to_the_interpreter(name, adapt_mass, monkeypatch):
    cp = E.compile_model(ZOO[name]())
    C, nw, ns = 150, 30, 20
    out, kernels = [], []
    for jit, W in [(0, 0), (1, 0), (1, 1), (1, 2), (1, 3), (1, 16)]:
        monkeypatch.setenv("FG_JIT", str(jit))
        if W: monkeypatch.setenv("FG_HMC_INTERP_WAVES", str(W))
        else: monkeypatch.delenv("FG_HMC_INTERP_WAVES", raising=False)
        eng = E.Engine(cp, C, seed=29, chain_offset=11)
        d = eng.device_alloc(ns * cp.d * C * 8)
        st = eng.hmc_run(E.hmc_config(n_leapfrog=5, adapt_mass=adapt_mass), ns, nw, d)
        kernels.append(eng.hmc_last_kernel())
        draws = eng.download(d, (ns, cp.d, C))
        eng.device_free(d)
        out.append((draws, eng.hmc_step_sizes(), eng.hmc_log_joint(), eng.get_values(), st.accept_rate, st.n_divergent,
                    eng.hmc_mass() if adapt_mass else None))
        eng.close()
    assert not kernels[0].startswith("k_hmc_jit_steps") and all(k.startswith("k_hmc_jit_steps W=") for k in kernels[1:]), kernels
    for o in out[1:]:
        for a, b in zip(out[0], o):
            assert (a is None and b is None) or np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
    assert np.isfinite(out[0][0]).all()


@pytest.mark.parametrize("name,adapt_mass", [("alldists", True), ("poisson_glm", False), ("hier_logsigma", True), ("logistic", False), ("coin", False), ("hier_scale", False),
                                             ("mixture", True), ("rand3", False), ("refmodel8", True), ("hier", False), ("linreg", True), ("ridge7", False)])
def test_jit_dense_mode_is_bit_identical_to_the_interpreter(name, adapt_mass, monkeypatch):
    """FG_GRAD_FD_DENSE -- grad_log_joint verbatim (hmc.rs:304-329): two WHOLE log-joints per coordinate and gradient -- through the unit
    compiled at run time (fg_jit_full_k: the whole program with the reads of coordinate k's slot replaced, one function per coordinate)
    against what FG_JIT=0 runs, W = 1 ... 16: the interpreter kernels for programs without a record stream and for stream programs with
    general records / option selects, the dense stream (k_hmc_stream_steps) for programs of fast-Normal records."""
    cp = E.compile_model(ZOO[name]())
    C, nw, ns = 130, 24, 12
    out, kernels = [], []
    # fused: the one-barrier gradient (whole coordinates per wave, a second copy of the site rows) forced on / off; None: the host's rule
    for jit, W, fused in [(0, 0, None), (1, 0, None), (1, 1, "1"), (1, 3, "0"), (1, 3, "1"), (1, 16, None), (1, 4, "1")]:
        monkeypatch.setenv("FG_JIT", str(jit))
        if W: monkeypatch.setenv("FG_HMC_INTERP_WAVES", str(W))
        else: monkeypatch.delenv("FG_HMC_INTERP_WAVES", raising=False)
        if fused is None: monkeypatch.delenv("FG_JIT_FUSED", raising=False)
        else: monkeypatch.setenv("FG_JIT_FUSED", fused)
        eng = E.Engine(cp, C, seed=31, chain_offset=2)
        d = eng.device_alloc(ns * cp.d * C * 8)
        st = eng.hmc_run(E.hmc_config(n_leapfrog=4, adapt_mass=adapt_mass, grad_mode=E.GRAD_FD_DENSE), ns, nw, d)
        kernels.append(eng.hmc_last_kernel())
        draws = eng.download(d, (ns, cp.d, C))
        eng.device_free(d)
        out.append((draws, eng.hmc_step_sizes(), eng.hmc_log_joint(), eng.get_values(), st.accept_rate, st.n_divergent,
                    eng.hmc_mass() if adapt_mass else None))
        eng.close()
    assert not kernels[0].startswith("k_hmc_jit_steps") and all(k.startswith("k_hmc_jit_steps W=") and "dense" in k for k in kernels[1:]), kernels
    for o in out[1:]:
        for a, b in zip(out[0], o):
            assert (a is None and b is None) or np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
    assert np.isfinite(out[0][0]).all()


@pytest.mark.parametrize("name,adapt_mass", [("hier_scale", True), ("mixture", False), ("linreg", True), ("refmodel8", False), ("hier", True), ("ridge7", False),
                                             ("rand1", False), ("rand3", True), ("rand4", False)])
def test_jit_hmc_is_bit_identical_to_the_stream_kernels(name, adapt_mass, monkeypatch):
    """Gradient-stream programs: the compiled form (the engine's choice for every one of them since round 4) against k_hmc_stream_steps
    (FG_JIT=0) -- the same arithmetic per coordinate, so every bit agrees."""
    cp = E.compile_model(ZOO[name]())
    assert cp.stream_records[0] > 0
    C, nw, ns = 150, 30, 20
    out, kernels = [], []
    monkeypatch.setenv("FG_HMC_LIN", "0")                                   # (ridge7 is a dense regression: its own kernel would take it)
    # tasks: the unit's task split as straight-line code per wave (default) / a task list in memory; fused: whole coordinates per wave and ONE barrier per
    # gradient (a second copy of the site rows) forced on / off (the host's choice otherwise)
    for jit, tasks, fused in ((0, 1, None), (1, 1, "1"), (1, 1, "0"), (1, 0, None)):
        monkeypatch.setenv("FG_JIT", str(jit))
        monkeypatch.setenv("FG_JIT_TASKS", str(tasks))
        if fused is None: monkeypatch.delenv("FG_JIT_FUSED", raising=False)
        else: monkeypatch.setenv("FG_JIT_FUSED", fused)
        eng = E.Engine(cp, C, seed=31, chain_offset=2)
        d = eng.device_alloc(ns * cp.d * C * 8)
        st = eng.hmc_run(E.hmc_config(n_leapfrog=6, adapt_mass=adapt_mass), ns, nw, d)
        kernels.append(eng.hmc_last_kernel())
        draws = eng.download(d, (ns, cp.d, C))
        eng.device_free(d)
        out.append((draws, eng.hmc_step_sizes(), eng.hmc_log_joint(), eng.get_values(), st.accept_rate, st.n_divergent, eng.hmc_mass() if adapt_mass else None))
        eng.close()
    assert kernels[0].startswith("k_hmc_stream_steps") and all(k.startswith("k_hmc_jit_steps") for k in kernels[1:]), kernels
    assert "one barrier per gradient" in kernels[1] and "one barrier" not in kernels[2] and "one barrier" not in kernels[3], kernels
    for o in out[1:]:
        for a, b in zip(out[0], o):
            assert (a is None and b is None) or np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def test_jit_is_not_used_where_a_faster_kernel_exists(monkeypatch):
    """Independent-sites programs and dense regressions keep their hand-written kernels (whole trajectories in registers -- in the sparse
    AND in the dense mode --, the observation-major gradient).  The dense mode of every other program is the compiled unit's since round 4
    (the whole program per coordinate: test_jit_dense_mode_is_bit_identical_to_the_interpreter)."""
    monkeypatch.delenv("FG_JIT", raising=False)
    for name, mode, jit in [("normal32", E.GRAD_FD_SPARSE, False), ("normal32", E.GRAD_FD_DENSE, False), ("ridge8", E.GRAD_FD_SPARSE, False),
                            ("poisson_glm", E.GRAD_FD_DENSE, True), ("refmodel8", E.GRAD_FD_DENSE, True), ("refmodel8", E.GRAD_FD_SPARSE, True)]:
        cp = E.compile_model(ZOO[name]())
        eng = E.Engine(cp, 64, seed=1)
        eng.hmc_init(E.hmc_config(grad_mode=mode, n_leapfrog=3), 2)
        eng.hmc_step(2)
        assert eng.hmc_last_kernel().startswith("k_hmc_jit_steps") == jit, (name, eng.hmc_last_kernel())
        eng.close()


@pytest.mark.parametrize("name,with_overrides", [("alldists", False), ("alldists", True), ("poisson_glm", False), ("hier_logsigma", True), ("logistic", False), ("coin", False)])
def test_jit_mh_is_bit_identical_to_the_interpreter(name, with_overrides, monkeypatch):
    """adaptive_mcmc_chain with the scoring run compiled at run time (eight generated statement segments shared by the waves of a
    tile, k_mh_jit_steps) against the one-wave interpreter kernel: recorded draws, final state, adapted scales, log-weights and accept
    counts, including steps whose proposal needs the model (undecided kinds, PriorResample overrides, computed Categorical tables)."""
    cp = E.compile_model(ZOO[name]())
    C, nw, ns = 150, 100, 40
    rec = list(range(cp.S))
    ov = None
    if with_overrides:
        ov = [None] * cp.S
        f64 = [j for j in range(cp.S) if cp.site_vtypes[j] == 0]
        ov[f64[0]] = (E.PROP_PRIOR_RESAMPLE, 0.0, 0.0)
        ov[f64[1]] = (E.PROP_GAUSSIAN, 0.0, 0.0)
    out, kernels = [], []
    monkeypatch.setenv("FG_MH_NOSTREAM_MW", "0")                           # the statement-segment kernel itself (the pipelined kernel: the next test)
    for jit, mw, W, occ in ((0, 0, 0, 0), (1, 1, 0, 0), (1, 1, 1, 2), (1, 1, 2, 4), (1, 1, 4, 2), (1, 1, 8, 4)):
        monkeypatch.setenv("FG_JIT", str(jit))
        monkeypatch.setenv("FG_HMC_INTERP_MW", str(mw))
        if W: monkeypatch.setenv("FG_MH_INTERP_WAVES", str(W)); monkeypatch.setenv("FG_MH_INTERP_OCC", str(occ))
        else: monkeypatch.delenv("FG_MH_INTERP_WAVES", raising=False); monkeypatch.delenv("FG_MH_INTERP_OCC", raising=False)
        eng = E.Engine(cp, C, seed=17, chain_offset=4)
        d = eng.device_alloc(max(1, ns * cp.S * C) * 8)
        st = eng.mh_run(ns, nw, ov, rec, d)
        kernels.append(eng.mh_last_kernel())
        draws = eng.download(d, (ns, cp.S, C), dtype=np.int64)
        eng.device_free(d)
        out.append((draws, eng.get_values(), eng.mh_scales(), eng.mh_log_weight(), st.accept_rate))
        eng.close()
    assert kernels[0] == "k_mh_steps W=1" and all(k.startswith("k_mh_jit_steps W=") for k in kernels[1:]), kernels
    for o in out[1:]:
        for a, b in zip(out[0], o):
            assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)

@pytest.mark.parametrize("name,with_overrides", [("alldists", False), ("alldists", True), ("poisson_glm", False), ("hier_logsigma", True), ("logistic", False), ("coin", False)])
def test_jit_mh_pipelined_kernel_for_programs_without_a_stream(name, with_overrides, monkeypatch):
    """Programs without a score stream on the pipelined multi-wave MH kernel (fg_mh_mw_body.h: random numbers of step t + 1 drawn
    while step t is scored, control wave, in-order sums) with every statement generated and the undecided-kind probe interpreted from
    the proposals that need the model (undecided kinds, PriorResample overrides, computed Categorical tables) interpreted from the
    target's own statement: the one-wave interpreter kernel's draws, state, scales, log-weights and accept counts for every W."""
    cp = E.compile_model(ZOO[name]())
    C, nw, ns = 150, 100, 40
    rec = list(range(cp.S))
    ov = None
    if with_overrides:
        ov = [None] * cp.S
        f64 = [j for j in range(cp.S) if cp.site_vtypes[j] == 0]
        ov[f64[0]] = (E.PROP_PRIOR_RESAMPLE, 0.0, 0.0)
        ov[f64[1]] = (E.PROP_GAUSSIAN, 0.0, 0.0)
    out, kernels = [], []
    for jit, Wv, split in ((0, 0, -1), (1, 0, -1), (1, 2, 0), (1, 4, 1), (1, 8, 0), (1, 16, 1)):
        monkeypatch.setenv("FG_JIT", str(jit))
        monkeypatch.setenv("FG_HMC_INTERP_MW", str(jit))
        if Wv: monkeypatch.setenv("FG_HMC_WAVES", str(Wv))
        else: monkeypatch.delenv("FG_HMC_WAVES", raising=False)
        if split >= 0: monkeypatch.setenv("FG_MH_SPLIT", str(split))
        else: monkeypatch.delenv("FG_MH_SPLIT", raising=False)
        eng = E.Engine(cp, C, seed=17, chain_offset=4)
        d = eng.device_alloc(max(1, ns * cp.S * C) * 8)
        st = eng.mh_run(ns, nw, ov, rec, d)
        kernels.append(eng.mh_last_kernel())
        draws = eng.download(d, (ns, cp.S, C), dtype=np.int64)
        eng.device_free(d)
        out.append((draws, eng.get_values(), eng.mh_scales(), eng.mh_log_weight(), st.accept_rate))
        eng.close()
    assert kernels[0] == "k_mh_steps W=1" and all(k.startswith("k_mh_mw_jit_steps W=") for k in kernels[1:]), kernels
    for o in out[1:]:
        for a, b in zip(out[0], o):
            assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)



@pytest.mark.parametrize("name", ["alldists", "poisson_glm", "hier_logsigma", "logistic", "hier_scale", "mixture", "linreg"])
def test_jit_smc_rejuvenation_is_identical_to_the_interpreter(name, monkeypatch):
    """adaptive_smc: the rejuvenation move's two scoring runs through the model compiled at run time (k_smc_jit_rejuv) against what FG_JIT=0
    runs -- the interpreter kernel (k_smc_rejuv<-1>) for a program without a score stream, the score-stream kernels (k_smc_rejuv<2 / 3>) for
    stream programs with general / option-select / linear-predictor records -- the same ladder, evidence, particles and weights, bit for bit."""
    cp = E.compile_model(ZOO[name]())
    out = []
    for jit in ("0", "1"):
        monkeypatch.setenv("FG_JIT", jit)
        eng = E.Engine(cp, 3000, seed=17)
        r = eng.smc_run(rejuvenation_steps=2, ess_threshold=0.5)
        out.append((r["betas"], r["log_evidence"], r["values"], r["weights"], r["n_model_runs"]))
        eng.close()
    assert len(out[0][0]) >= 2
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.mark.parametrize("name", ["refmodel8", "hier"])
def test_jit_smc_of_a_large_population_is_identical(name, monkeypatch):
    """From 262 144 particles on adaptive_smc builds the compiled unit for the prior draw and then rejuvenates through it whatever the
    program's records (fast-Normal score streams included): against FG_JIT=0 (k_prior_init, k_smc_rejuv<0>) the same ladder, evidence,
    particles and weights, bit for bit."""
    cp = E.compile_model(ZOO[name]())
    out = []
    for jit in ("0", "1"):
        monkeypatch.setenv("FG_JIT", jit)
        eng = E.Engine(cp, 1 << 18, seed=19)
        r = eng.smc_run(rejuvenation_steps=2, ess_threshold=0.5)
        out.append((r["betas"], r["log_evidence"], r["values"], r["weights"], r["n_model_runs"]))
        eng.close()
    assert len(out[0][0]) >= 2
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.mark.parametrize("name", ["alldists", "poisson_glm", "hier_logsigma", "logistic", "coin", "mixture", "rand0", "rand3"])
def test_jit_prior_draw_is_identical_to_the_interpreter(name, monkeypatch):
    """run(PriorHandler) through the model compiled at run time (k_prior_jit: every sample statement draws through the same fg_sample_dist
    from the same stream, then scores) against the interpreter kernel k_prior_init: the same cells of every site of every chain and the
    same three accumulators, bit for bit; likewise the re-score of a session's log-joint after set_values (k_log_joint_jit).  Programs
    with a record stream only use the compiled draw once the unit exists: an HMC step in the compiled mode builds it first."""
    cp = E.compile_model(ZOO[name]())
    out = []
    for jit in ("0", "1"):
        monkeypatch.setenv("FG_JIT", jit)
        eng = E.Engine(cp, 700, seed=29, chain_offset=3)
        if jit == "1" and cp.d > 0:                          # build the unit: the step-size search of a session runs on the compiled kernel here
            eng.hmc_init(E.hmc_config(n_leapfrog=2), 0)       # (few tiles: the compiled form is the engine's choice for stream programs too)
        acc = eng.prior_init(5)
        vals = eng.get_values()
        out.append((acc, vals))
        eng.close()
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][0], out[1][0], equal_nan=True)


@pytest.mark.parametrize("name", ["alldists", "logistic", "poisson_glm", "hier_logsigma", "hier_scale", "mixture", "coin", "rand3"])
def test_analytic_gradients_of_any_program_match_the_finite_difference(name):
    """FG_GRAD_ANALYTIC beyond Normal force terms (north_star: "finite-difference (and where available analytic) gradients"; opt-in):
    the forward-mode derivative of every sub-program in the unit compiled at run time (Gen::ins_ad: the 17 distributions' partial
    derivatives -- fg_dlogpdf, checked on the host against difference quotients in tests/cpp/test_dlogpdf.cpp -- and the chain rule
    through the expression opcodes) against the reference's central difference: the same trajectories to the finite difference's own
    noise.  Fixed step size, a few transitions; a chain whose accept test sits on a knife edge may part ways."""
    cp = E.compile_model(ZOO[name]())
    C, ns = 256, 4
    out = {}
    for mode in (E.GRAD_FD_SPARSE, E.GRAD_ANALYTIC):
        eng = E.Engine(cp, C, seed=23)
        d = eng.device_alloc(ns * cp.d * C * 8)
        st = eng.hmc_run(E.hmc_config(grad_mode=mode, n_leapfrog=5, init_step_size=0.02), ns, 0, d)
        out[mode] = (eng.download(d, (ns, cp.d, C)), st.accept_rate, eng.hmc_last_kernel())
        eng.device_free(d)
        eng.close()
    fd, an = out[E.GRAD_FD_SPARSE], out[E.GRAD_ANALYTIC]
    assert an[2].startswith("k_hmc_jit_steps"), an[2]
    bad = (~np.isclose(fd[0], an[0], rtol=2e-5, atol=2e-6)).any(axis=(0, 1))
    assert bad.sum() <= 2, (name, int(bad.sum()), np.abs(fd[0] - an[0]).max())
    assert np.isfinite(an[0]).all() and abs(fd[1] - an[1]) < 0.02 and an[1] > 0.3


def test_jit_rolls_plates_and_scores_long_programs_directly(monkeypatch):
    """A plate of observations (statements that differ only in their constants) becomes one loop over a constant table in the
    generated code; a program with more statements than LDS has term rows runs MH with the in-order accumulators on one wave
    (k_mh_jit_steps' direct mode).  Logistic regression with 300 observations: HMC and MH against the interpreter kernels, bit for bit."""
    from fugue_amd import workloads as W
    cp = E.compile_model(W.logistic_regression(*W.classification_data(300)[:2]))
    C = 130
    out = []
    monkeypatch.setenv("FG_MH_NOSTREAM_MW", "0")                           # (the pipelined kernel has term rows for this program: the segment kernel's direct mode is what is tested)
    for jit in (0, 1):
        monkeypatch.setenv("FG_JIT", str(jit))
        eng = E.Engine(cp, C, seed=5)
        eng.hmc_init(E.hmc_config(n_leapfrog=4, init_step_size=0.02), 6)
        eng.hmc_step(10)
        kh = eng.hmc_last_kernel()
        v = eng.get_values(); lj = eng.hmc_log_joint(); eps = eng.hmc_step_sizes()
        eng.mh_init(30)
        eng.mh_step(50)
        km = eng.mh_last_kernel()
        out.append((v, lj, eps, eng.get_values(), eng.mh_scales(), eng.mh_log_weight(), kh, km))
        eng.close()
    assert out[1][6].startswith("k_hmc_jit_steps") and "in-order accumulators" in out[1][7], out[1][6:]
    assert not out[0][6].startswith("k_hmc_jit") and not out[0][7].startswith("k_mh_jit"), out[0][6:]
    for a, b in zip(out[0][:6], out[1][:6]):
        assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("seed", range(16))
def test_jit_matches_the_interpreter_on_random_expression_programs(seed, monkeypatch):
    """tests/random_models.py::random_expression_program: random expression trees over every interpreter opcode and distribution
    family, plates, selects, factors -- HMC and MH through the compiled kernels and through the interpreter kernels, bit for bit
    (NaN and -inf included: chains that leave a family's support must do so identically)."""
    from tests.random_models import random_expression_program
    cp = E.compile_model(random_expression_program(seed))
    C = 100
    out = []
    for jit in (0, 1):
        monkeypatch.setenv("FG_JIT", str(jit))
        eng = E.Engine(cp, C, seed=40 + seed)
        eng.prior_init()
        eng.hmc_init(E.hmc_config(n_leapfrog=4, init_step_size=0.01), 5)
        eng.hmc_step(10)
        kh = eng.hmc_last_kernel()
        v = eng.get_values(); lj = eng.hmc_log_joint(); eps = eng.hmc_step_sizes()
        eng.mh_init(20)
        eng.mh_step(40)
        out.append((v, lj, eps, eng.get_values(), eng.mh_scales(), eng.mh_log_weight(), kh, eng.mh_last_kernel()))
        eng.close()
    assert out[1][6].startswith("k_hmc_jit_steps") and out[1][7].startswith(("k_mh_jit_steps", "k_mh_mw_jit_steps")), out[1][6:]
    for a, b in zip(out[0][:6], out[1][:6]):
        assert np.array_equal(a, b, equal_nan=True)


def test_a_failed_compilation_falls_back_to_the_interpreter_kernels(monkeypatch, tmp_path):
    """When hiprtc refuses the generated unit (FG_JIT_BREAK appends an #error; the same path as a missing libhiprtc), the engine stays on
    the interpreter kernels -- on the GPU, with the same results -- and says which kernel ran."""
    cp = E.compile_model(ZOO["poisson_glm"]())
    monkeypatch.setenv("FG_JIT_CACHE", str(tmp_path))                    # an empty cache: nothing compiled earlier can be picked up
    out = []
    for broken in (False, True):
        if broken: monkeypatch.setenv("FG_JIT_BREAK", "1")
        eng = E.Engine(cp, 100, seed=3)
        eng.hmc_init(E.hmc_config(n_leapfrog=4), 5); eng.hmc_step(8)
        kh = eng.hmc_last_kernel(); v = eng.get_values()
        eng.mh_init(10); eng.mh_step(20)
        out.append((v, eng.get_values(), eng.mh_scales(), kh, eng.mh_last_kernel()))
        eng.close()
    assert out[0][3].startswith("k_hmc_jit_steps") and out[0][4].startswith(("k_mh_jit_steps", "k_mh_mw_jit_steps")), out[0][3:]
    assert out[1][3].startswith("k_hmc_interp_mw_steps") and out[1][4].startswith("k_mh_interp_mw_steps"), out[1][3:]
    for a, b in zip(out[0][:3], out[1][:3]):
        assert np.array_equal(a, b, equal_nan=True)


def test_code_objects_are_cached_on_disk_and_verified(tmp_path):
    """FG_JIT_CACHE: a second process finds the compiled code object of the same program (no second compilation: the run is faster),
    and a cache file that was truncated or belongs to another source text is ignored, not trusted."""
    import glob, os, subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "run.py"
    script.write_text(f"""
import sys, time
sys.path.insert(0, {root!r})
from fugue_amd import engine as E
from tests.models import ZOO
cp = E.compile_model(ZOO["poisson_glm"]())
eng = E.Engine(cp, 64, seed=1)
t0 = time.perf_counter()
eng.hmc_init(E.hmc_config(n_leapfrog=3), 2); eng.hmc_step(3); eng.synchronize()
print(eng.hmc_last_kernel(), "|", time.perf_counter() - t0, "|", eng.get_values().tobytes().hex()[:64])
""")
    env = dict(os.environ, FG_JIT_CACHE=str(tmp_path / "cache"), FG_JIT="1")

    def run():
        r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        k, t, v = [x.strip() for x in r.stdout.strip().splitlines()[-1].split("|")]
        assert k.startswith("k_hmc_jit_steps"), k
        return float(t), v
    t1, v1 = run()
    files = glob.glob(str(tmp_path / "cache" / "*.hsaco"))
    assert len(files) == 1
    t2, v2 = run()
    assert v2 == v1 and t2 < t1                                   # found on disk: no compilation
    size = os.path.getsize(files[0])
    with open(files[0], "r+b") as f:                              # damage the file: cut its tail off, then flip a byte of the stored source text
        f.truncate(size // 2)
    t3, v3 = run()
    assert v3 == v1 and os.path.getsize(files[0]) == size         # ignored, recompiled, rewritten
    with open(files[0], "r+b") as f:
        f.seek(100); b = f.read(1); f.seek(100); f.write(bytes([b[0] ^ 1]))
    t4, v4 = run()
    assert v4 == v1


def test_compilation_inside_a_process_that_runs_on_a_bundled_hip_runtime(tmp_path):
    """A host process that imported PyTorch runs on the wheel's own libamdhip64 / libhiprtc / libamd_comgr, whose hiprtc produced code
    objects that did not run for these units ("invalid kernel file").  There the library compiles with the system ROCm's hipcc in a
    child process instead: same results as the interpreter, in a fresh cache directory."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "run.py"
    script.write_text(f"""
import sys, os
import torch                                   # first: the process binds the wheel's HIP runtime
sys.path.insert(0, {root!r})
from fugue_amd import engine as E
from tests.models import ZOO
cp = E.compile_model(ZOO["hier_logsigma"]())
out = []
for jit in ("0", "1"):
    os.environ["FG_JIT"] = jit
    eng = E.Engine(cp, 100, seed=1)
    eng.hmc_init(E.hmc_config(n_leapfrog=3), 3); eng.hmc_step(5)
    kh = eng.hmc_last_kernel(); v = eng.get_values().tobytes()
    eng.mh_init(10); eng.mh_step(20); eng.synchronize()
    out.append((kh, eng.mh_last_kernel(), v, eng.get_values().tobytes()))
    eng.close()
assert out[1][0].startswith("k_hmc_jit_steps") and out[1][1].startswith(("k_mh_jit_steps", "k_mh_mw_jit_steps")), out[1][:2]
assert out[0][2] == out[1][2] and out[0][3] == out[1][3]
print("ok")
""")
    env = dict(os.environ, FG_JIT_CACHE=str(tmp_path / "cache"))
    env.pop("FG_JIT_COMPILER", None)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-2000:])


def test_jit_mh_segments_with_rolled_plates(monkeypatch):
    """Sixty observations: each of the eight statement segments of the compiled MH kernel holds a run of isomorphic statements long
    enough to roll into a loop whose iteration r writes term row (base + r) -- against the interpreter kernels, bit for bit."""
    from fugue_amd import workloads as W
    cp = E.compile_model(W.logistic_regression(*W.classification_data(60)[:2]))
    out = []
    monkeypatch.setenv("FG_MH_NOSTREAM_MW", "0")
    for jit, W_ in ((0, 0), (1, 1), (1, 2), (1, 8)):
        monkeypatch.setenv("FG_JIT", str(jit))
        if W_: monkeypatch.setenv("FG_MH_INTERP_WAVES", str(W_))
        else: monkeypatch.delenv("FG_MH_INTERP_WAVES", raising=False)
        eng = E.Engine(cp, 140, seed=9)
        eng.prior_init()
        eng.mh_init(40); eng.mh_step(80)
        out.append((eng.get_values(), eng.mh_scales(), eng.mh_log_weight(), eng.mh_last_kernel()))
        eng.close()
    assert all(o[3].startswith("k_mh_jit_steps W=") and "in-order" not in o[3] for o in out[1:]), [o[3] for o in out]
    for o in out[1:]:
        for a, b in zip(out[0][:3], o[:3]):
            assert np.array_equal(a, b, equal_nan=True)
