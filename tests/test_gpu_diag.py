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
