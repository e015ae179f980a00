"""The reference's golden log-density fixtures and edge cases through the HIP path.

Every entry of tests/golden/reference_kats.json (the reference's own known-answer tests) and
tests/golden/logpdf_vectors.json (vectors generated from the reference's tests/gen_refs.py and scipy), plus the edge
cases the reference tests (distribution.rs:925-945,1155-1160,1630-1640,2291-2353,2525-2592;
tests/f_dist_distributions.rs:153-332), is evaluated on the device by every code path that computes a log-density:

  const / interpreter : all parameters compile-time constants -> the hoisted path of `fg_logpdf` (fg_hoist constants,
                        FG_F_INVALID, FG_OP_NORMAL_FAST for Normals) in the interpreter kernel (`fg_log_joint`);
  site  / interpreter : every parameter fed from a sample site -> the non-hoisted path (guards evaluated per lane);
  const / stream      : the statement as a 64-byte score-stream record (FG_G_GEN hoisted, fast Normal, FG_G_CATC) --
                        what the HMC endpoint, the MH model run and SMC rejuvenation evaluate (`fg_log_joint_stream`);
  site  / stream      : FG_G_GEN records with FG_G_GEN_PkSLOT operands.

Tolerance: the fixture's own (1e-9 absolute for the reference KATs, 1e-9 relative for the vectors); +-inf exact.
"""
import json
import math
import os

import numpy as np
import pytest

from fugue_amd import engine as E
from fugue_amd import model as M

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
KATS = json.load(open(os.path.join(G, "reference_kats.json")))
VECS = json.load(open(os.path.join(G, "logpdf_vectors.json")))
INT_DISTS = {"Bernoulli", "Categorical", "Binomial", "Poisson", "DiscreteUniform"}
I64_MIN, I64_MAX = -(1 << 63), (1 << 63) - 1


def _f(v):
    return {"inf": math.inf, "-inf": -math.inf, "nan": math.nan}[v] if isinstance(v, str) else v


def _entries():
    """(dist, params, x, expected, abs tol, rel tol, label)"""
    out = []
    for k in KATS["logpdf"]:
        out.append((k["dist"], [_f(p) for p in k["params"]], _f(k["x"]), _f(k["expected"]), k["tol"], 0.0, "kat:" + k["src"]))
    for k in KATS["discrete_uniform_i64"]:
        out.append(("DiscreteUniform", [k["lo"], k["hi"]], k["x"], _f(k["expected"]), max(k["tol"], 1e-12), 0.0, "kat:" + k["src"]))
    for part in ("gen_refs", "scipy"):
        for v in VECS[part]:
            out.append((v["dist"], list(v["params"]), v["x"], v["expected"], 1e-9, 1e-9, part))
    return out


# Edge cases of the reference's log_prob bodies; `None` = take the expected value from the oracle (which the KATs pin).
LN2 = math.log(2.0)
EDGES = [
    # Beta endpoints (distribution.rs:925-945; tests/f_dist_distributions.rs:198-250)
    ("Beta", [0.5, 0.5], 0.0, math.inf), ("Beta", [0.5, 0.5], 1.0, math.inf), ("Beta", [0.5, 2.0], 0.0, math.inf),
    ("Beta", [2.0, 0.5], 1.0, math.inf), ("Beta", [2.0, 2.0], 0.0, -math.inf), ("Beta", [2.0, 2.0], 1.0, -math.inf),
    ("Beta", [1.0, 5.0], 0.0, math.log(5.0)), ("Beta", [3.0, 1.0], 1.0, math.log(3.0)), ("Beta", [1.0, 1.0], 0.0, 0.0),
    ("Beta", [2.0, 3.0], -0.1, -math.inf), ("Beta", [2.0, 3.0], 1.1, -math.inf), ("Beta", [2.0, 3.0], math.nan, -math.inf),
    ("Beta", [0.5, 0.5], 1e-100, 113.98452476385289), ("Beta", [2.0, 2.0], 1e-100, -228.46674983017652),
    # Weibull at and below zero (distribution.rs:1630-1640)
    ("Weibull", [0.5, 1.0], 0.0, math.inf), ("Weibull", [1.0, 2.0], 0.0, -LN2), ("Weibull", [2.0, 1.0], 0.0, -math.inf),
    ("Weibull", [1.5, 2.0], -1.0, -math.inf), ("Weibull", [1.5, 2.0], math.inf, -math.inf),
    # Binomial degenerate p (distribution.rs:1155-1160)
    ("Binomial", [5, 0.0], 0, 0.0), ("Binomial", [5, 0.0], 2, -math.inf), ("Binomial", [5, 1.0], 5, 0.0), ("Binomial", [5, 1.0], 4, -math.inf),
    ("Binomial", [5, 0.3], 6, -math.inf), ("Binomial", [5, 0.3], -1, -math.inf), ("Binomial", [0, 0.3], 0, None),
    # Bernoulli degenerate p (tests/f_dist_distributions.rs:257-293)
    ("Bernoulli", [0.0], 0, 0.0), ("Bernoulli", [0.0], 1, -math.inf), ("Bernoulli", [1.0], 1, 0.0), ("Bernoulli", [1.0], 0, -math.inf),
    # Poisson: lambda > 700 and k = 0 shortcut (distribution.rs:1246-1248), and the formula just below it
    ("Poisson", [800.0], 0, -800.0), ("Poisson", [700.5], 0, -700.5), ("Poisson", [700.0], 0, -700.0), ("Poisson", [800.0], 3, None),
    ("Poisson", [3.0], 0, -3.0), ("Poisson", [3.0], -1, -math.inf),
    # DiscreteUniform full range and support (distribution.rs:2291-2353, 2525-2592)
    ("DiscreteUniform", [I64_MIN, I64_MAX], 0, -64.0 * LN2), ("DiscreteUniform", [I64_MIN, I64_MAX], I64_MIN, -64.0 * LN2),
    ("DiscreteUniform", [I64_MIN, I64_MAX - 1], I64_MAX, -math.inf), ("DiscreteUniform", [I64_MIN + 1, I64_MAX], I64_MIN, -math.inf),
    ("DiscreteUniform", [I64_MIN + 1, I64_MAX], 5, None), ("DiscreteUniform", [3, 3], 3, 0.0), ("DiscreteUniform", [3, 3], 4, -math.inf),
    # Uniform: half-open support
    ("Uniform", [-2.0, 2.0], -2.0, -math.log(4.0)), ("Uniform", [-2.0, 2.0], 2.0, -math.inf), ("Uniform", [-2.0, 2.0], math.nan, -math.inf),
    ("Uniform", [-2.0, 2.0], math.nextafter(2.0, 0.0), -math.log(4.0)),
    # Categorical: out of bounds, zero probability (tests/f_dist_distributions.rs:318-332)
    ("Categorical", [0.2, 0.8], 2, -math.inf), ("Categorical", [0.2, 0.8], -1, -math.inf), ("Categorical", [0.0, 1.0], 0, -math.inf),
    ("Categorical", [0.0, 1.0], 1, 0.0), ("Categorical", [1.0], 0, 0.0),
    # removed overflow guards (distribution.rs:2116-2134; tests/f_dist_distributions.rs:153-191)
    ("Gamma", [2.0, 1.0], 800.0, -793.315388272332), ("Normal", [0.0, 0.001], 0.05, -1244.0111832542225),
    ("Normal", [0.0, 1.0], 40.0, -800.9189385332047), ("LogNormal", [0.0, 0.001], 1.05, -1184.3000332584572),
    ("Exponential", [2.0], 400.0, -799.3068528194401),
    # support boundaries of the positive families
    ("Gamma", [2.0, 1.0], 0.0, -math.inf), ("Gamma", [2.0, 1.0], -1.0, -math.inf), ("LogNormal", [0.0, 1.0], 0.0, -math.inf),
    ("Exponential", [2.0], -0.5, -math.inf), ("Exponential", [2.0], 0.0, LN2), ("ChiSquared", [4.0], 0.0, -math.inf),
    ("InverseGamma", [3.0, 2.0], 0.0, -math.inf), ("InverseGamma", [3.0, 2.0], -1.0, -math.inf),
    ("StudentT", [3.0, 1.0, 2.0], 1e200, None), ("Cauchy", [0.0, 1.0], 1e200, None), ("Laplace", [0.0, 1.0], 1e300, None),
]
# a non-finite value is -inf for every continuous family (first guard after the parameter guards)
CONT = {"Normal": [0.3, 0.7], "Uniform": [-1.0, 2.0], "LogNormal": [0.1, 0.9], "Exponential": [1.3], "Beta": [2.0, 3.0],
        "Gamma": [2.0, 1.5], "StudentT": [4.0, 0.5, 2.0], "Cauchy": [0.5, 0.25], "Laplace": [-0.5, 4.0],
        "Weibull": [1.5, 2.0], "ChiSquared": [4.0], "InverseGamma": [3.0, 2.0]}
for _d, _p in CONT.items():
    for _x in (math.nan, math.inf, -math.inf):
        EDGES.append((_d, _p, _x, -math.inf))
# invalid parameters -> -inf whatever the value (guard order of every log_prob body)
INVALID = [("Normal", [0, 0], 0.0), ("Normal", [0, -1], 0.0), ("Normal", [math.nan, 1], 0.0), ("Normal", [0, math.inf], 0.0),
           ("Uniform", [1, 1], 1.0), ("Uniform", [2, 1], 1.5), ("LogNormal", [0, 0], 1.0), ("LogNormal", [math.inf, 1], 1.0),
           ("Exponential", [0], 1.0), ("Exponential", [-1], 1.0), ("Bernoulli", [1.5], 1), ("Bernoulli", [-0.1], 0), ("Bernoulli", [math.nan], 0),
           ("Beta", [0, 1], 0.5), ("Beta", [1, -1], 0.5), ("Gamma", [1, 0], 1.0), ("Gamma", [0, 1], 1.0), ("Binomial", [5, 1.5], 1),
           ("Binomial", [5, math.nan], 1), ("Poisson", [0], 1), ("Poisson", [-2], 0), ("StudentT", [0, 0, 1], 0.0), ("StudentT", [3, 0, 0], 0.0),
           ("StudentT", [3, math.nan, 1], 0.0), ("Cauchy", [0, 0], 0.0), ("Laplace", [0, 0], 0.0), ("Laplace", [math.nan, 1], 0.0),
           ("Weibull", [0, 1], 1.0), ("Weibull", [1, 0], 1.0), ("ChiSquared", [0], 1.0), ("ChiSquared", [-1], 1.0),
           ("InverseGamma", [1, 0], 1.0), ("InverseGamma", [0, 1], 1.0)]
for _d, _p, _x in INVALID:
    EDGES.append((_d, [float(v) for v in _p], _x, -math.inf))


def _is_intlike(v):
    return isinstance(v, (int, np.integer)) and not isinstance(v, bool)


def _streamable(dist, params, form):
    """Does the compiler have a score-stream record form for this statement? (fg_program.cpp: FAST / GEN / CATC)"""
    if dist == "DiscreteUniform":
        return False
    if dist == "Categorical":
        return form == "const"
    if dist == "Binomial" and form == "const":       # hoisted Binomial keeps three constants: interpreter only
        return False
    return True


def _sat(v: float) -> int:
    """Rust `as i64` (fg_f2i_sat)"""
    if v != v:
        return 0
    if v >= 9223372036854775807.0:
        return I64_MAX
    if v <= -9223372036854775808.0:
        return I64_MIN
    return int(v)


def _site_form_ok(dist, params):
    if dist == "DiscreteUniform":                    # site-fed bounds travel as f64: only exactly representable ones
        return all(_sat(float(p)) == int(p) for p in params)
    return True


def _run_batch(batch, form, want_stream):
    """batch: list of (dist, params, x).  Returns (interpreter log-densities, stream log-densities or None), one per entry."""
    P = M.Program()
    v_handles, v_stmt = [], []
    p_handles = []
    for i, (dist, params, x) in enumerate(batch):
        if form == "const":
            if dist == "DiscreteUniform" and all(_is_intlike(p) for p in params):
                d = M.Dist(dist, [M.as_expr(float(p)) for p in params], (int(params[0]), int(params[1])))
            else:
                d = M.Dist(dist, [M.as_expr(float(p)) for p in params])       # no constructor validation: invalid constants must score -inf
            p_handles.append([])
        else:
            ps = [P.sample(M.addr(f"p{k}", i), M.Normal(0.0, 1.0)) for k in range(len(params))]
            p_handles.append([e.a for e in ps])
            d = M.Dist(dist, ps)
        v_stmt.append(len(P.stmts))
        v_handles.append(P.sample(M.addr("v", i), d).a)
    cp = E.compile_model(P)
    C = 67                                                # two waves, the second one ragged
    cells = np.zeros((cp.S, C), dtype=np.int64)

    def put(handle, value):
        j = cp.site_of_handle(handle)
        if cp.site_vtypes[j] == 0:
            cells[j, :] = np.array([float(value)], dtype=np.float64).view(np.int64)[0]
        else:
            cells[j, :] = int(value)
    for i, (dist, params, x) in enumerate(batch):
        put(v_handles[i], x)
        for h, p in zip(p_handles[i], params):
            put(h, float(p))
    eng = E.Engine(cp, C, seed=1)
    eng.set_values(cells)
    _, logp = eng.log_joint(want_logp=True)
    interp = []
    for i in range(len(batch)):
        col = logp[cp.site_of_handle(v_handles[i])]
        assert np.array_equal(col, np.full(C, col[0]), equal_nan=True), "lanes disagree"
        interp.append(float(col[0]))
    stream = None
    if want_stream:
        assert cp.stream_records[1] == len(P.stmts), (form, "expected a score stream for this batch", cp.stream_records)
        _, rec = eng.log_joint_stream(want_records=True)
        stream = []
        for i in range(len(batch)):
            col = rec[v_stmt[i]]
            assert np.array_equal(col, np.full(C, col[0]), equal_nan=True), "lanes disagree"
            stream.append(float(col[0]))
    eng.close()
    return interp, stream


def _check(got, exp, atol, rtol, what):
    if math.isinf(exp) or math.isnan(exp):
        assert got == exp or (math.isnan(exp) and math.isnan(got)), (what, got, exp)
    else:
        assert math.isfinite(got) and abs(got - exp) <= atol + rtol * max(1.0, abs(exp)), (what, got, exp)


def _sweep(entries, form):
    n_interp = n_stream = 0
    for want_stream in (True, False):
        sel = [e for e in entries if _streamable(e[0], e[1], form) == want_stream and (form == "const" or _site_form_ok(e[0], e[1]))]
        chunk = 96 if form == "const" else 48
        for b in range(0, len(sel), chunk):
            part = sel[b:b + chunk]
            interp, stream = _run_batch([(d, p, x) for d, p, x, *_ in part], form, want_stream)
            for k, (d, p, x, exp, atol, rtol, label) in enumerate(part):
                _check(interp[k], exp, atol, rtol, (form, "interpreter", d, p, x, label))
                n_interp += 1
                if stream is not None:
                    _check(stream[k], exp, atol, rtol, (form, "stream", d, p, x, label))
                    n_stream += 1
    return n_interp, n_stream


@pytest.mark.parametrize("form", ["const", "site"])
def test_golden_fixtures_on_device(form):
    """reference_kats.json + logpdf_vectors.json through the interpreter and the score-stream records."""
    ents = _entries()
    n_interp, n_stream = _sweep(ents, form)
    assert n_interp >= len(ents) - 2 and n_stream >= 300, (n_interp, n_stream)


@pytest.mark.parametrize("form", ["const", "site"])
def test_reference_edge_cases_on_device(oracle, form):
    """Endpoints, degenerate parameters, non-finite values, invalid parameters: literal expectations where the reference
    asserts one, the (KAT-pinned) oracle otherwise -- and the oracle must agree with every literal."""
    ents = []
    for d, p, x, exp in EDGES:
        if d == "DiscreteUniform":
            o = oracle.logpdf_discrete_uniform(x, int(p[0]), int(p[1]))
        else:
            o = oracle.logpdf(d, x, p)
        if exp is None:
            exp = o
        else:
            _check(o, exp, 1e-9, 0.0, ("oracle vs literal", d, p, x))
        if d == "Categorical" and form == "const" and abs(sum(p) - 1.0) > 1e-6:
            continue
        ents.append((d, p, x, exp, 1e-9, 1e-12, "edge"))
    n_interp, n_stream = _sweep(ents, form)
    assert n_interp >= len(ents) - 8 and n_stream >= 100, (n_interp, n_stream)


def test_gradient_stream_general_records_see_the_same_density(oracle):
    """FG_G_GEN records of the fused finite-difference gradient stream call the same fg_logpdf at q +- h: on a model made
    of leaf-operand statements the sparse (stream) force equals the dense (whole-program) force and the oracle's, also
    where a perturbed coordinate sits on a support boundary (Gamma / Exponential / Beta sites near 0 -> -inf -> not ok)."""
    P = M.Program()
    g = P.sample(M.addr("g"), M.Gamma(2.0, 2.0))
    b = P.sample(M.addr("b"), M.Beta(2.0, 3.0))
    mu = P.sample(M.addr("mu"), M.Normal(0.0, 2.0))
    P.observe(M.addr("y"), M.Normal(mu, g), 0.4)
    P.observe(M.addr("f"), M.Bernoulli(b), 1.0)
    P.observe(M.addr("w"), M.Weibull(1.5, g), 0.8)
    cp, om = E.compile_model(P), oracle.OracleModel(P)
    assert cp.stream_records[0] > 0 and cp.stream_records[2] == 2
    C = 8
    vals = {"g": [1.0, 1e-6, 5e-6, 0.3, 2.0, 1e-5 + 1e-12, 0.7, 3.0], "b": [0.5, 0.5, 1e-6, 1.0 - 1e-6, 0.2, 0.9, 5e-6, 0.4],
            "mu": [0.0, 1.0, -1.0, 0.3, 2.0, -2.0, 0.1, 0.5]}
    cells = np.zeros((cp.S, C), dtype=np.int64)
    for j, name in enumerate(cp.site_names):
        cells[j] = np.array(vals[name], dtype=np.float64).view(np.int64)
    eng = E.Engine(cp, C, seed=1)
    eng.set_values(cells)
    gd, okd = eng.hmc_grad(1e-5, E.GRAD_FD_DENSE)
    gs, oks = eng.hmc_grad(1e-5, E.GRAD_FD_SPARSE)
    for c in range(C):
        q = np.ascontiguousarray(cells[om.f64_sites, c]).view(np.float64)
        og, ook = om.grad_log_joint(cells[:, c], q)
        assert ook == bool(okd[c]) == bool(oks[c]), c
        fin = np.isfinite(og)
        assert np.array_equal(np.isfinite(gd[:, c]), fin) and np.array_equal(np.isfinite(gs[:, c]), fin), c
        lj = abs(om.log_joint_at(cells[:, c], q))
        tol = 5e-6 * (1.0 + (lj if np.isfinite(lj) else 0.0))
        assert np.allclose(gd[fin, c], og[fin], rtol=1e-7, atol=tol) and np.allclose(gs[fin, c], og[fin], rtol=1e-7, atol=tol), c
    eng.close()
