"""GPU parity: per-chain moment / autocovariance kernels feeding split-R-hat and multichain ESS,
on draws produced by the HMC kernel, against the oracle's diagnostics on the same draws."""
import numpy as np
import pytest

from fugue_amd import diagnostics as D
from fugue_amd import engine as E
from fugue_amd import workloads as W
from tests.diag_helpers import NumpyMoments

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ns", [61, 200])
def test_chain_diagnostics_on_hmc_draws(oracle, ns):
    cp = E.compile_model(W.normal_sites(5))
    C, nw = 200, 60
    eng = E.Engine(cp, C, seed=2)
    d_draws = eng.device_alloc(ns * cp.d * C * 8)
    eng.hmc_run(E.hmc_config(grad_mode=E.GRAD_FD_SPARSE, n_leapfrog=4), ns, nw, d_draws)
    draws = eng.download(d_draws, (ns, cp.d, C))
    prov = D.EngineMoments(eng, d_draws, ns, cp.d)
    ref = NumpyMoments(draws)
    np.testing.assert_allclose(prov.moments(), ref.moments(), rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(prov.autocov_sums(0, 40), ref.autocov_sums(0, 40), rtol=1e-9, atol=1e-10)
    cd = D.ChainDiagnostics(prov)
    for i in range(cp.d):
        ch = np.ascontiguousarray(draws[:, i, :].T)
        assert cd.split_rhat()[i] == pytest.approx(oracle.split_rhat(ch), rel=1e-10)
        assert cd.classic_rhat()[i] == pytest.approx(oracle.classic_rhat(ch), rel=1e-10)
        assert cd.ess()[i] == pytest.approx(oracle.ess_multichain(ch), rel=1e-8)
    assert (cd.split_rhat() < 1.05).all()
    prov.close()


@pytest.mark.parametrize("exchange", [E.DIAG_REDUCE, E.DIAG_GATHER])
def test_native_rhat_ess_entry_point_and_geweke(oracle, exchange):
    """fg_diag_rhat_ess (moments + lag sums on the device, combination in C++) and fg_diag_geweke against the oracle, in both
    exchange modes (all-reduces of chain sums formed on the device; all-gather of every chain's moments); with a one-rank RCCL
    communicator the same call goes through ncclAllReduce / ncclAllGather and must not change a bit."""
    cp = E.compile_model(W.normal_sites(4))
    C, ns, nw = 150, 240, 60
    eng = E.Engine(cp, C, seed=6)
    d_draws = eng.device_alloc(ns * cp.d * C * 8)
    eng.hmc_run(E.hmc_config(grad_mode=E.GRAD_FD_SPARSE, n_leapfrog=3), ns, nw, d_draws)
    draws = eng.download(d_draws, (ns, cp.d, C))
    r = eng.diag_rhat_ess(d_draws, ns, cp.d, exchange=exchange)
    assert r["chains"] == C and r["exchange_bytes"] == 0
    for i in range(cp.d):
        ch = np.ascontiguousarray(draws[:, i, :].T)
        assert r["r_hat"][i] == pytest.approx(oracle.split_rhat(ch), rel=1e-10)
        assert r["ess"][i] == pytest.approx(oracle.ess_multichain(ch), rel=1e-8)
        s = oracle.summarize(ch)
        assert r["mean"][i] == pytest.approx(s["mean"], rel=1e-11, abs=1e-12) and r["std"][i] == pytest.approx(s["std"], rel=1e-10)
    z = eng.diag_geweke(d_draws, ns, cp.d)
    for i in range(cp.d):
        for c in (0, 7, 149):
            assert z[i, c] == pytest.approx(oracle.geweke(np.ascontiguousarray(draws[:, i, c])), rel=1e-9, abs=1e-12, nan_ok=True)
    assert np.isnan(eng.diag_geweke(d_draws, 15, cp.d)).all()                 # n < 20 (mcmc_utils.rs:356-358)
    comm = eng.comm_init(1, 0, E.comm_unique_id())
    try:
        r1 = eng.diag_rhat_ess(d_draws, ns, cp.d, comm)
    finally:
        E.comm_destroy(comm)
    for k in ("r_hat", "ess", "mean", "std"):
        assert np.array_equal(r[k], r1[k])
    lag_bytes = r1["exchange_bytes"] - (8 * cp.d * 8 if exchange == E.DIAG_REDUCE else 6 * cp.d * C * 8)
    assert lag_bytes > 0 and lag_bytes % (32 * cp.d * 8) == 0                  # what the rank put into collectives: O(d) in the reduce mode
    eng.device_free(d_draws)


def test_quantiles_are_selected_on_the_device(oracle):
    """fg_diag_quantiles == summarize_f64_parameter's rule sorted[round((len - 1) p)] (diagnostics.rs:355-371): on HMC draws
    against the oracle's orc_summarize, and on adversarial values (ties, both zeros, infinities, denormals, one repeated value)
    against a host sort -- the element itself, bit for bit."""
    cp = E.compile_model(W.normal_sites(3))
    C, ns = 130, 97
    eng = E.Engine(cp, C, seed=8)
    d_draws = eng.device_alloc(ns * cp.d * C * 8)
    eng.hmc_run(E.hmc_config(n_leapfrog=3), ns, 30, d_draws)
    draws = eng.download(d_draws, (ns, cp.d, C))
    q = eng.diag_quantiles(d_draws, ns, cp.d)
    for i in range(cp.d):
        s = oracle.summarize(np.ascontiguousarray(draws[:, i, :].T))
        assert [q[i, k] for k in range(5)] == [s["q2.5"], s["q25"], s["q50"], s["q75"], s["q97.5"]]
        assert q[i].tolist() == [D.quantiles_f64(draws[:, i, :])[k] for k in ("2.5%", "25%", "50%", "75%", "97.5%")]
    rng = np.random.default_rng(4)
    x = rng.standard_normal((ns, cp.d, C))
    x[:, 0] = np.round(x[:, 0] * 3) / 3                                       # heavy ties
    x[::7, 0] = 0.0; x[1::7, 0] = -0.0
    x[:, 1] = 2.5                                                             # a constant coordinate
    x[:5, 2] = np.inf; x[5:9, 2] = -np.inf; x[9:12, 2] = 5e-324; x[12:15, 2] = -5e-324
    buf = np.ascontiguousarray(x)
    E._check(E.lib().fg_device_upload(eng.h, d_draws, buf.ctypes.data, buf.nbytes))
    probs = (0.0, 0.001, 0.025, 0.5, 0.975, 0.999, 1.0)
    q = eng.diag_quantiles(d_draws, ns, cp.d, probs)
    for i in range(cp.d):
        v = np.sort(x[:, i, :].ravel())
        want = [v[int(np.floor((len(v) - 1) * p + 0.5))] for p in probs]
        assert np.array_equal(q[i], np.array(want))                          # -0.0 == 0.0 here: the reference's comparison cannot tell them apart either
    comm = eng.comm_init(1, 0, E.comm_unique_id())
    try:
        assert np.array_equal(eng.diag_quantiles(d_draws, ns, cp.d, probs, comm), q)
    finally:
        E.comm_destroy(comm)
    with pytest.raises(E.EngineError):
        eng.diag_quantiles(d_draws, ns, cp.d, (1.5,))
    eng.device_free(d_draws)
