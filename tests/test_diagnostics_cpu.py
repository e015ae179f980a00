"""Host-side diagnostics logic (fugue_amd/diagnostics.py) against the oracle's restatement of
diagnostics.rs / mcmc_utils.rs, the reference's own diagnostics tests restated
(tests/f_mcmc_diagnostics.rs:30-139, src/inference/mcmc_utils.rs:472-570), and the N > 1 path
under a world_size-2 gloo group."""
import os
import subprocess
import sys

import numpy as np
import pytest

from fugue_amd import diagnostics as D
from tests.diag_helpers import NumpyMoments, ar1

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _chains(draws, i):                     # [n][d][C] -> [m][n] for the oracle
    return np.ascontiguousarray(draws[:, i, :].T)


@pytest.mark.parametrize("exchange", ["reduce", "gather"])
@pytest.mark.parametrize("n", [7, 50, 401])
def test_rhat_and_ess_match_oracle(oracle, n, exchange):
    rng = np.random.default_rng(n)
    draws = np.stack([ar1(rng, n, 12, 0.6), rng.standard_normal((n, 12)) * 3 + 1, ar1(rng, n, 12, 0.95)], axis=1)
    cd = D.ChainDiagnostics(NumpyMoments(draws), exchange=exchange)
    for i in range(3):
        ch = _chains(draws, i)
        assert cd.split_rhat()[i] == pytest.approx(oracle.split_rhat(ch), rel=1e-11)
        assert cd.classic_rhat()[i] == pytest.approx(oracle.classic_rhat(ch), rel=1e-11)
        assert cd.ess()[i] == pytest.approx(oracle.ess_multichain(ch), rel=1e-9)
        s = oracle.summarize(ch)
        assert cd.pooled_mean()[i] == pytest.approx(s["mean"], rel=1e-11, abs=1e-12)
        assert cd.pooled_std()[i] == pytest.approx(s["std"], rel=1e-10)
        q = D.quantiles_f64(draws[:, i, :])
        for k in ("2.5%", "25%", "50%", "75%", "97.5%"):
            assert q[k] == s["q" + k[:-1]]
        assert D.geweke_diagnostic(ch[0]) == pytest.approx(oracle.geweke(ch[0]), rel=1e-9, nan_ok=True)


def test_reference_diagnostics_behaviour(oracle):
    rng = np.random.default_rng(1)
    # iid chains: split-R-hat < 1.01; multichain ESS ~ m*n  (tests/f_mcmc_diagnostics.rs:30-139)
    x = rng.standard_normal((1000, 1, 4))
    cd = D.ChainDiagnostics(NumpyMoments(x))
    assert cd.split_rhat()[0] < 1.01 and 0.7 * 4000 < cd.ess()[0] <= 4000
    # a drift shared by all chains: classic R-hat blind (<1.01), split R-hat sees it (>1.1)
    y = rng.standard_normal((1000, 1, 4)) * 0.1 + np.linspace(0, 3, 1000)[:, None, None]
    cd = D.ChainDiagnostics(NumpyMoments(y))
    assert cd.classic_rhat()[0] < 1.01 and cd.split_rhat()[0] > 1.1
    # scale invariance to 1e-9 and the AR(1) answer (1-phi)/(1+phi) within 15 % (mcmc_utils.rs:472-528)
    z = ar1(rng, 20000, 1, 0.9)[:, None, :]
    e1 = D.ChainDiagnostics(NumpyMoments(z)).ess()[0]
    e2 = D.ChainDiagnostics(NumpyMoments(z * 1000.0)).ess()[0]
    assert e1 == pytest.approx(e2, rel=1e-9)
    assert e1 / 20000 == pytest.approx(0.1 / 1.9, rel=0.15)
    # constant chains: every draw counts (mcmc_utils.rs:276-279); < 4 draws: total count (:259-261)
    assert D.ChainDiagnostics(NumpyMoments(np.ones((50, 1, 3)))).ess()[0] == 150.0
    assert D.ChainDiagnostics(NumpyMoments(rng.standard_normal((3, 1, 5)))).ess()[0] == 15.0
    # single chain: R-hat needs two (diagnostics.rs:263-265) -> split halves of one chain still give 2
    assert oracle.classic_rhat(np.ones((1, 10))) == 1.0


@pytest.mark.parametrize("exchange", ["reduce", "gather"])
def test_distributed_diagnostics_world_size_2_gloo(oracle, tmp_path, exchange):
    """The N > 1 path: two gloo ranks each own half of the chains, exchange only chain sums / lag sums (reduce: all-reduces of
    O(d) doubles; gather: every chain's moments) and must reproduce the single-process (and the oracle's) R-hat and ESS."""
    rng = np.random.default_rng(3)
    draws = np.stack([ar1(rng, 300, 16, 0.7), rng.standard_normal((300, 16)) + 2.0], axis=1)
    np.save(tmp_path / "draws.npy", draws)
    script = tmp_path / "worker.py"
    script.write_text(f'''
import os, sys, json
sys.path.insert(0, {ROOT!r})
import numpy as np, torch.distributed as dist
from fugue_amd import diagnostics as D
from tests.diag_helpers import NumpyMoments
dist.init_process_group(backend="gloo")
r, w = dist.get_rank(), dist.get_world_size()
x = np.load({str(tmp_path / "draws.npy")!r})
C = x.shape[2] // w
cd = D.ChainDiagnostics(NumpyMoments(x[:, :, r * C:(r + 1) * C]), exchange={exchange!r})
out = dict(split=cd.split_rhat().tolist(), classic=cd.classic_rhat().tolist(), ess=cd.ess().tolist(), m=cd.m, mean=cd.pooled_mean().tolist(),
           std=cd.pooled_std().tolist(), bytes=cd.exchange_bytes)
if r == 0:
    json.dump(out, open({str(tmp_path / "out.json")!r}, "w"))
dist.destroy_process_group()
''')
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29517" if exchange == "reduce" else "29519", str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    out = json.load(open(tmp_path / "out.json"))
    assert out["m"] == 16
    lag_chunks = out["bytes"] - (2 * 8 * 8 if exchange == "reduce" else 2 * 6 * 8 * 8)     # d = 2: 6 d + 2 d doubles, or [d][6][8 chains]
    assert lag_chunks > 0 and lag_chunks % (2 * 32 * 8) == 0                             # + 32 d doubles per chunk of lags
    single = D.ChainDiagnostics(NumpyMoments(draws), exchange=exchange)
    for i in range(2):
        ch = _chains(draws, i)
        assert out["split"][i] == pytest.approx(oracle.split_rhat(ch), rel=1e-11)
        assert out["classic"][i] == pytest.approx(oracle.classic_rhat(ch), rel=1e-11)
        assert out["ess"][i] == pytest.approx(oracle.ess_multichain(ch), rel=1e-9)
        assert out["ess"][i] == pytest.approx(single.ess()[i], rel=1e-12)
        s = oracle.summarize(ch)
        assert out["mean"][i] == pytest.approx(s["mean"], rel=1e-11, abs=1e-12) and out["std"][i] == pytest.approx(s["std"], rel=1e-10)


def test_combine_is_native_and_needs_no_gpu():
    """The R-hat / Geyer-ESS combination runs in the library (fg_diag_combine, C++): same numbers as the oracle on ragged
    sizes, NaN / degenerate inputs follow the reference's rules (diagnostics.rs:262-270, mcmc_utils.rs:259-279)."""
    from fugue_amd import engine as E
    rng = np.random.default_rng(5)
    x = np.stack([ar1(rng, 120, 6, 0.5), np.ones((120, 6))], axis=1)
    nm = NumpyMoments(x)
    r = E.diag_combine(nm.moments(), 120, nm.autocov_sums)
    assert r["ess"][1] == 720.0 and np.isnan(r["r_hat"][1])                 # constant chains: every draw counts; W = 0 -> NaN
    one = NumpyMoments(x[:, :1, :1])
    r1 = E.diag_combine(one.moments(), 120, one.autocov_sums)
    assert np.isfinite(r1["r_hat"][0])                                      # one chain still splits into two halves
    tiny = NumpyMoments(rng.standard_normal((1, 1, 3)))
    assert E.diag_combine(tiny.moments(), 1, tiny.autocov_sums)["r_hat"][0] == pytest.approx(D.ChainDiagnostics(tiny).classic_rhat()[0], nan_ok=True)
