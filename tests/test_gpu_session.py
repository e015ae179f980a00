"""The incremental step(n) / values_since protocol (crates/fugue-wasm/src/mh.rs:44-283, hmc.rs:46-156) over the engine:
fugue_amd.session.MhSession / HmcSession."""
import numpy as np
import pytest

import fugue_amd as F
from fugue_amd import engine as E
from fugue_amd.session import HmcSession, MhSession

pytestmark = pytest.mark.gpu
addr = F.addr

MODEL = lambda: F.sample(addr("mu"), F.Normal(0.0, 2.0)).bind(lambda mu: F.observe(addr("y"), F.Normal(mu, 1.0), 2.5).map(lambda _: mu))
DSL = 'let mu <- sample(addr!("mu"), Normal(0.0, 2.0)); observe(addr!("y"), Normal(mu, 1.0), data[0]); pure(mu)'


def test_mh_session_protocol_and_history():
    s = MhSession(MODEL, n_chains=64, seed=7, max_history=400)
    assert s.n_chains() == 64 and s.site_names() == ["mu"] and s.total_steps() == 0
    assert np.isnan(s.acceptance_rate()) and s.values_since("mu", 0, 0).size == 0
    first = s.current_values("mu").copy()
    assert s.step(100) == 100 and s.total_steps() == 100
    v = s.values_since("mu", 3, 0)
    assert v.shape == (100,) and np.isfinite(v).all() and v[-1] == s.current_values("mu")[3]
    assert s.values_since("mu", 3, 60).shape == (40,) and np.array_equal(s.values_since("mu", 3, 60), v[60:])
    assert s.values_since("nope", 0, 0).size == 0 and s.values_since("mu", 999, 0).size == 0
    assert (s.current_values("mu") != first).any() and np.isfinite(s.log_weights()).all()
    # recording runs while the chains adapt: the same engine without the switch records nothing before the warmup ends
    s.step(400)                                                 # history cap 400: the oldest half is dropped in blocks (mh.rs:120-127)
    assert s.total_steps() == 500 and s.hist.buf.shape[0] <= 400 and s.hist.dropped > 0
    tail = s.values_since("mu", 3, 450)
    assert tail.shape == (50,) and tail[-1] == s.current_values("mu")[3]
    m, sd, rhat, ess = s.summary("mu")
    assert abs(m - 2.0) < 0.1 and abs(sd - np.sqrt(0.8)) < 0.1 and 0.9 < rhat < 1.1 and ess > 100
    assert 0.2 < s.acceptance_rate() < 0.8 and 0.9 < s.r_hat("mu", 100) < 1.2 and s.ess("mu") > 100
    s.set_value(5, "mu", 9.5)
    assert s.current_values("mu")[5] == 9.5
    s.step(50)
    assert abs(s.current_values("mu")[5] - 2.0) < 4.0           # the chain walks back to the posterior from the planted value
    s.close()


def test_mh_session_matches_a_plain_run_of_the_same_steps():
    """The session is the engine's own chain: the states it keeps are the states of mh_step with recording switched on."""
    s = MhSession(MODEL, n_chains=32, seed=11)
    s.step(30); s.step(20)
    cp = E.compile_model(MODEL)
    eng = E.Engine(cp, 32, seed=11)
    eng.mh_init(2 ** 31 - 1)
    eng.mh_set_recording(True)
    buf = eng.device_alloc(50 * 32 * 8)
    eng.mh_step(50, [0], buf)
    ref = eng.download(buf, (50, 1, 32), dtype=np.int64)
    assert np.array_equal(ref[:, 0, :], s.hist.buf[:, 0, :])
    eng.close(); s.close()


def test_hmc_session_protocol():
    s = HmcSession(MODEL, n_chains=128, seed=3, n_warmup=50, n_leapfrog=8)
    assert s.site_names() == ["mu"] and s.is_warming_up()
    assert s.step(30) == 30 and s.values("mu").shape == (30, 128) and s.is_warming_up()          # every transition's state is kept, warmup included (hmc.rs:104-115, 159-164)
    assert s.step(40) == 70 and not s.is_warming_up() and s.values("mu").shape == (70, 128)
    assert np.array_equal(s.values("mu")[-1], s.eng.get_values()[0].view(np.float64))             # the last kept state is the chains' current one
    rec = s.step_recorded(chain=5)
    assert rec["n_points"] == 9 and rec["positions"].shape == (9, 1) and np.isfinite(rec["hamiltonians"]).all()
    assert s.values("mu").shape == (71, 128)                    # the recorded transition's state joins the history (hmc.rs:83-101)
    assert np.array_equal(s.values("mu")[-1], s.eng.get_values()[0].view(np.float64))
    s.set_n_leapfrog(12)
    s.step(200)
    x = s.values("mu")[71:]
    assert x.shape == (200, 128) and abs(x.mean() - 2.0) < 0.05 and s.ess("mu") > 1000
    assert (s.step_size() > 0.0).all()
    s.close()


def test_sessions_take_the_model_language():
    s = MhSession(DSL, "[2.5]", n_chains=64, seed=1)
    s.step(300)
    assert abs(s.summary("mu")[0] - 2.0) < 0.15
    s.close()
