"""Pins the CPU oracle against the reference's own known-answer tests (tests/golden/
reference_kats.json, harvested from src/core/distribution.rs, tests/f_dist_distributions.rs,
tests/f_dist_numerical.rs) and against vectors generated from the reference's
tests/gen_refs.py and scipy (tests/golden/logpdf_vectors.json)."""
import json
import math
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")
KATS = json.load(open(os.path.join(G, "reference_kats.json")))
VECS = json.load(open(os.path.join(G, "logpdf_vectors.json")))


def _f(v):
    return {"inf": math.inf, "-inf": -math.inf}.get(v, v) if isinstance(v, str) else float(v)


def _check(got, exp, tol):
    exp = _f(exp)
    if math.isinf(exp):
        assert got == exp
    else:
        assert math.isfinite(got) and abs(got - exp) <= tol, (got, exp)


@pytest.mark.parametrize("k", KATS["logpdf"], ids=lambda k: f"{k['dist']}{k['params']}@{k['x']}")
def test_logpdf_reference_kats(oracle, k):
    _check(oracle.logpdf(k["dist"], k["x"], k["params"]), k["expected"], k["tol"])


@pytest.mark.parametrize("k", KATS["discrete_uniform_i64"], ids=lambda k: f"DU[{k['lo']},{k['hi']}]@{k['x']}")
def test_discrete_uniform_full_range(oracle, k):
    _check(oracle.logpdf_discrete_uniform(k["x"], k["lo"], k["hi"]), k["expected"], k["tol"])


@pytest.mark.parametrize("part", ["gen_refs", "scipy"])
def test_logpdf_vectors(oracle, part):
    assert len(VECS[part]) > 100
    for v in VECS[part]:
        got = oracle.logpdf(v["dist"], v["x"], v["params"])
        assert abs(got - v["expected"]) <= 1e-9 * max(1.0, abs(v["expected"])), (v, got)


def test_invalid_parameters_give_neg_inf(oracle):
    # guard order of every log_prob body (distribution.rs:189...1932)
    for dist, params, x in [("Normal", [0, 0], 0.0), ("Normal", [0, -1], 0.0), ("Normal", [math.nan, 1], 0.0),
                            ("Normal", [0, 1], math.inf), ("Uniform", [1, 1], 1.0), ("LogNormal", [0, 0], 1.0),
                            ("Exponential", [0], 1.0), ("Bernoulli", [1.5], 1), ("Beta", [0, 1], 0.5),
                            ("Gamma", [1, 0], 1.0), ("Binomial", [5, 1.5], 1), ("Poisson", [0], 1),
                            ("StudentT", [0, 0, 1], 0.0), ("Cauchy", [0, 0], 0.0), ("Laplace", [0, 0], 0.0),
                            ("Weibull", [0, 1], 1.0), ("ChiSquared", [0], 1.0), ("InverseGamma", [1, 0], 1.0),
                            ("DiscreteUniform", [3, 1], 2)]:
        assert oracle.logpdf(dist, x, params) == -math.inf, (dist, params, x)
    assert oracle.logpdf("Poisson", 0, [800.0]) == -800.0          # distribution.rs:1246-1248


def test_log_sum_exp_kats(oracle):
    for k in KATS["numerical"]["lse"]:
        _check(oracle.log_sum_exp([_f(v) for v in k["x"]]), k["expected"], 1e-12)
    for k in KATS["numerical"]["softmax"]:
        np.testing.assert_allclose(oracle.normalize_log_probs(k["x"]), k["expected"], rtol=0, atol=1e-15)
    for k in KATS["numerical"]["log1p_exp"]:
        _check(oracle.log1p_exp(k["x"]), k["expected"], k["tol"])
    assert oracle.safe_ln(0.0) == -math.inf and oracle.safe_ln(1.0) == 0.0


def test_philox_known_answer(oracle):
    # Random123 KAT for philox4x32-10: all-zero and all-ones counters/keys
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
