"""GPU half of the model-language tests (crates/fugue-wasm/src/dsl.rs:1149-1328): a DSL-built site
program must score, prior-sample and run inference exactly like the hand-built mirror program, which
the other parity tests tie to the oracle; here the oracle is consulted directly as well."""
import numpy as np
import pytest

from fugue_amd import engine as E
from tests.dsl_models import PAIRS
from tests.models import f64_values_for

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(PAIRS))
def test_dsl_program_scores_and_samples_like_mirror_and_oracle(oracle, name):
    src, data, mirror = PAIRS[name]
    prog = mirror()
    cp, ref, om = E.CompiledProgram.from_dsl(src, data), E.compile_model(prog), oracle.OracleModel(prog)
    C = 130
    a, b = E.Engine(cp, C, seed=9), E.Engine(ref, C, seed=9)
    acc_a, acc_b = a.prior_init(), b.prior_init()
    assert np.array_equal(a.get_values(), b.get_values())              # bit-identical prior draws
    assert np.array_equal(acc_a, acc_b, equal_nan=True)
    cells = f64_values_for(om, np.random.default_rng(4), C)
    a.set_values(cells), b.set_values(cells)
    la, pa = a.log_joint(want_logp=True)
    lb, pb = b.log_joint(want_logp=True)
    assert np.array_equal(la, lb, equal_nan=True) and np.array_equal(pa, pb, equal_nan=True)
    for c in range(0, C, 13):                                          # and the oracle, 1e-12 (libm ulps)
        oacc, _ = om.run_score(cells[:, c])
        assert np.allclose(la[:, c], oacc, rtol=1e-12, atol=1e-12, equal_nan=True)


def test_coin_model_posterior(oracle):
    """adaptive_mcmc_chain on the interpreted coin model recovers Beta(2+7, 2+3): mean 9/14 within 0.03
    (dsl.rs:1159-1187, there with one chain of 4000 draws; here 256 chains of 400)."""
    cp = E.CompiledProgram.from_dsl(PAIRS["coin"][0], PAIRS["coin"][1])
    C, ns = 256, 400
    eng = E.Engine(cp, C, seed=11)
    d = eng.device_alloc(ns * C * 8)
    eng.mh_run(ns, 300, None, [0], d)
    p = eng.download(d, (ns, 1, C))
    eng.device_free(d)
    assert ((p > 0) & (p < 1)).all()
    assert abs(p.mean() - 9.0 / 14.0) < 0.03 / 4


def test_regression_model_with_named_arrays():
    """dsl.rs:1189-1212 checks 0.4 < mean(a) < 1.4; the exact conjugate posterior mean of the slope is
    sum(xy) / (sum(x^2) + 0.8^2/2.5^2) = 8.4 / 10.1024."""
    cp = E.CompiledProgram.from_dsl(PAIRS["regression"][0], PAIRS["regression"][1])
    C, ns = 512, 200
    eng = E.Engine(cp, C, seed=3)
    d = eng.device_alloc(ns * cp.d * C * 8)
    eng.hmc_run(E.hmc_config(), ns, 300, d)
    draws = eng.download(d, (ns, cp.d, C))
    eng.device_free(d)
    assert cp.site_names == ["a", "b"]
    assert abs(draws[:, 0, :].mean() - 8.4 / 10.1024) < 0.02


def test_invalid_params_kill_weight_not_process():
    """dsl.rs:1258-1277: the prior draw still happens, total log-weight is -inf."""
    cp = E.CompiledProgram.from_dsl('let mu <- sample(addr!("mu"), Normal(0.0, -1.0)); pure(mu)')
    eng = E.Engine(cp, 64, seed=0)
    acc = eng.prior_init()
    assert np.isfinite(eng.get_values().view(np.float64)).all()
    assert (acc.sum(axis=0) == -np.inf).all()
    assert cp.warnings


def test_discrete_sites_sample_and_observe():
    """dsl.rs:1299-1327: u64 / usize sites exist and the total weight is finite."""
    cp = E.CompiledProgram.from_dsl(PAIRS["discrete"][0])
    eng = E.Engine(cp, 256, seed=2)
    acc = eng.prior_init()
    v = eng.get_values()
    assert cp.site_names == ["k", "z"] and cp.site_vtypes[0] != 0 and cp.site_vtypes[1] != 0
    assert (v[0] >= 0).all() and set(np.unique(v[1])) <= {0, 1}
    assert np.isfinite(acc.sum(axis=0)).all()
