"""Checkpoint / resume and trajectory recording: fg_state_export / fg_state_import, fg_hmc_step_recorded and the
HmcSession setters / getters (hmc.rs:741-787, 811-817, 1058-1087)."""
import numpy as np
import pytest

from fugue_amd import engine as E
from tests.models import ZOO

pytestmark = pytest.mark.gpu


def _hmc_draws(eng, n, d, C):
    buf = eng.device_alloc(max(1, n * d * C) * 8)
    eng.hmc_step(n, buf)
    out = eng.download(buf, (n, d, C))
    eng.device_free(buf)
    return out


@pytest.mark.parametrize("name,adapt_mass,mode", [("normal32", True, E.GRAD_FD_SPARSE), ("refmodel8", True, E.GRAD_FD_SPARSE),
                                                  ("hier_scale", False, E.GRAD_FD_SPARSE), ("readme", False, E.GRAD_FD_DENSE), ("ridge7", True, E.GRAD_ANALYTIC)])
def test_hmc_state_round_trip_is_bit_identical(name, adapt_mass, mode):
    """run 20; export; NEW engine; import; run 20  ==  run 40, bit for bit -- across the warmup / mass-reset / frozen
    step-size boundaries (n_warmup = 30: the reset falls at 15, sampling starts at 30)."""
    cp = E.compile_model(ZOO[name]())
    C, nw = 130, 30
    cfg = E.hmc_config(grad_mode=mode, n_leapfrog=6, adapt_mass=adapt_mass)
    a = E.Engine(cp, C, seed=9, chain_offset=11)
    a.hmc_init(cfg, nw)
    ref = _hmc_draws(a, 40, cp.d, C)          # rows of warmup transitions stay zero: only the last 10 are draws
    ref_vals, ref_eps, ref_lj = a.get_values(), a.hmc_step_sizes(), a.hmc_log_joint()
    for cut in (10, 15, 20, 30, 33):
        b = E.Engine(cp, C, seed=9, chain_offset=11)
        b.hmc_init(cfg, nw)
        b.hmc_step(cut)
        assert b.hmc_iterations() == cut and b.hmc_is_warming_up() == (cut < nw)
        blob = b.state_export()
        b.close()
        c = E.Engine(cp, C, seed=1234, chain_offset=0)     # another seed / offset: the blob carries the stream key
        c.state_import(blob)
        assert c.hmc_iterations() == cut
        rest = _hmc_draws(c, 40 - cut, cp.d, C)
        n_draw_rows = 40 - max(cut, nw)
        assert np.array_equal(rest[:n_draw_rows], ref[:10][10 - n_draw_rows:]) if n_draw_rows else True
        assert np.array_equal(c.get_values(), ref_vals)
        assert np.array_equal(c.hmc_step_sizes(), ref_eps) and np.array_equal(c.hmc_log_joint(), ref_lj)
        if adapt_mass:
            assert np.array_equal(c.hmc_mass(), a.hmc_mass())
        c.close()


@pytest.mark.parametrize("name", ["refmodel8", "mixture", "alldists"])
def test_mh_state_round_trip_is_bit_identical(name):
    cp = E.compile_model(ZOO[name]())
    C, nw, total = 100, 40, 70
    rec = list(range(cp.S))

    def run(eng, n):
        buf = eng.device_alloc(max(1, n * cp.S * C) * 8)
        eng.mh_step(n, rec, buf)
        out = eng.download(buf, (n, cp.S, C), dtype=np.int64)
        eng.device_free(buf)
        return out
    a = E.Engine(cp, C, seed=5, chain_offset=2)
    a.mh_init(nw)
    ref = run(a, total)
    for cut in (13, 40, 55):
        b = E.Engine(cp, C, seed=5, chain_offset=2)
        b.mh_init(nw)
        b.mh_step(cut)
        blob = b.state_export()
        c = E.Engine(cp, C, seed=77)
        c.state_import(blob)
        rest = run(c, total - cut)
        n_rows = total - max(cut, nw)
        assert np.array_equal(rest[:n_rows], ref[:total - nw][(total - nw) - n_rows:])
        assert np.array_equal(c.get_values(), a.get_values()) and np.array_equal(c.mh_scales(), a.mh_scales())
        assert np.array_equal(c.mh_log_weight(), a.mh_log_weight())


def test_state_import_refuses_a_foreign_blob():
    cp1, cp2 = E.compile_model(ZOO["readme"]()), E.compile_model(ZOO["normal32"]())
    a = E.Engine(cp1, 64, seed=1)
    a.hmc_init(E.hmc_config(), 5)
    blob = a.state_export()
    b = E.Engine(cp2, 64, seed=1)
    with pytest.raises(E.EngineError):
        b.state_import(blob)
    with pytest.raises(E.EngineError):
        E.Engine(cp1, 65, seed=1).state_import(blob)
    with pytest.raises(E.EngineError):
        E.Engine(cp1, 64, seed=1).state_import(b"garbage" * 40)


def test_state_import_validates_the_layout_before_touching_the_engine():
    """A blob whose header lies about its sections -- a short `bytes`, flipped flag bits, a truncated buffer -- is refused
    before any copy runs (the copies would otherwise read past the caller's buffer), and a refused import leaves the engine
    exactly as it was: same state, same session, still steppable and bit-identical to an engine that never saw the blob."""
    import struct
    cp = E.compile_model(ZOO["normal32"]())
    C = 64
    a = E.Engine(cp, C, seed=1)
    a.hmc_init(E.hmc_config(), 5)
    a.hmc_step(3)
    blob = a.state_export()
    # header layout (fg_state.hip FgStateHeader): magic u64, version u32, flags u32 at byte 12; `bytes` u64 is its last field
    flags = struct.unpack_from("<I", blob, 12)[0]
    assert flags == 1
    hdr = len(blob) - (cp.S * C * 8 + 9 * C * 8)
    bad = []
    bad.append(blob[:len(blob) - 8])                                                       # truncated buffer, honest header
    bad.append(blob[:hdr - 8] + struct.pack("<Q", len(blob) - 4096) + blob[hdr:])          # header claims fewer bytes than its sections need
    bad.append(blob[:12] + struct.pack("<I", flags | 2) + blob[16:])                       # mass section flagged but absent
    bad.append(blob[:12] + struct.pack("<I", flags | 4 | 8) + blob[16:])                   # MH sections flagged but absent
    bad.append(blob[:12] + struct.pack("<I", flags | 64) + blob[16:])                      # unknown flag bit
    bad.append(blob[:12] + struct.pack("<I", 2) + blob[16:])                               # mass adaptation without an HMC section
    b, ref = E.Engine(cp, C, seed=9), E.Engine(cp, C, seed=9)
    for e_ in (b, ref):
        e_.hmc_init(E.hmc_config(), 4)
        e_.hmc_step(2)
    for x in bad:
        with pytest.raises(E.EngineError):
            b.state_import(x)
        assert b.hmc_iterations() == 2 and np.array_equal(b.get_values(), ref.get_values())
    b.hmc_step(6); ref.hmc_step(6)
    assert np.array_equal(b.get_values(), ref.get_values()) and np.array_equal(b.hmc_step_sizes(), ref.hmc_step_sizes())
    b.state_import(blob)                                                                   # the honest blob still imports
    assert b.hmc_iterations() == 3 and np.array_equal(b.get_values(), a.get_values())


def test_mh_resume_keeps_recording_while_adapting():
    """An MhSession-style engine records every step, adapting or not (fg_mh_set_recording); export / import must carry that
    switch, or the resumed engine writes no rows while it is still warming up."""
    cp = E.compile_model(ZOO["refmodel8"]())
    C, nw = 64, 1 << 20
    rec = list(range(cp.S))

    def run(eng, n):
        buf = eng.device_alloc(n * cp.S * C * 8)
        eng.upload_zeros(buf, n * cp.S * C * 8) if hasattr(eng, "upload_zeros") else None
        eng.mh_step(n, rec, buf)
        out = eng.download(buf, (n, cp.S, C), dtype=np.int64)
        eng.device_free(buf)
        return out
    a = E.Engine(cp, C, seed=5)
    a.mh_init(nw); a.mh_set_recording(True)
    ref = run(a, 30)
    b = E.Engine(cp, C, seed=5)
    b.mh_init(nw); b.mh_set_recording(True)
    first = run(b, 12)
    c = E.Engine(cp, C, seed=123)
    c.state_import(b.state_export())
    rest = run(c, 18)
    assert np.array_equal(first, ref[:12]) and np.array_equal(rest, ref[12:])


@pytest.mark.parametrize("name,mode", [("readme", E.GRAD_FD_DENSE), ("normal32", E.GRAD_FD_SPARSE), ("refmodel8", E.GRAD_FD_SPARSE), ("alldists", E.GRAD_FD_DENSE)])
def test_step_recorded_is_rng_neutral_and_matches_the_oracle(oracle, name, mode):
    """hmc.rs:1058-1087: a recorded trajectory has L + 1 points with finite Hamiltonians, recording does not perturb the
    chain, and each point equals the oracle's leapfrog from the same state and momentum."""
    prog = ZOO[name]()
    cp, om = E.compile_model(prog), oracle.OracleModel(prog)
    C, L = 96, 12
    cfg = E.hmc_config(n_leapfrog=L, init_step_size=0.05, grad_mode=mode)
    rec, plain = E.Engine(cp, C, seed=3), E.Engine(cp, C, seed=3)
    rec.hmc_init(cfg, 0); plain.hmc_init(cfg, 0)
    ids = [0, 5, 63, 64, 95]
    for it in range(6):
        before = rec.get_values()
        traj, ham, npts = rec.hmc_step_recorded(ids, L)
        plain.hmc_step(1)
        assert np.array_equal(rec.get_values(), plain.get_values())          # recording consumes no randomness
        assert np.array_equal(rec.hmc_log_joint(), plain.hmc_log_joint())
        for k, c in enumerate(ids):
            assert 1 <= npts[k] <= L + 1
            assert np.isfinite(ham[k, :npts[k]]).all()
            q0 = np.ascontiguousarray(before[om.f64_sites, c]).view(np.float64)
            assert np.array_equal(traj[k, 0], q0)                             # the start point is the current position
            p0, _ = oracle.hmc_momentum(3, c, it, cp.d)
            qo, po, div = om.leapfrog(before[:, c], q0.copy(), p0.copy(), 0.05, L)
            if not div:
                assert npts[k] == L + 1
                assert np.allclose(traj[k, L], qo, rtol=1e-6, atol=1e-8)      # FD-force tolerance of the transition tests
                lj_end = om.log_joint_at(before[:, c], qo)
                assert abs(ham[k, L] - (-lj_end + 0.5 * np.dot(po, po))) < 1e-5 * (1 + abs(lj_end))
    assert rec.hmc_iterations() == 6


def test_session_setters():
    """set_n_leapfrog (hmc.rs:751-753), is_warming_up / iterations (:780-787), and set_step_size leaving warmup AND mass
    adaptation (hmc.rs:741-747, 877-908): the mass matrix must not change after the step size was pinned."""
    cp = E.compile_model(ZOO["normal32"]())
    eng = E.Engine(cp, 64, seed=4)
    eng.hmc_init(E.hmc_config(grad_mode=E.GRAD_FD_SPARSE, adapt_mass=True), 40)
    assert eng.hmc_is_warming_up() and eng.hmc_iterations() == 0
    eng.hmc_step(5)
    eng.hmc_set_step_size(0.07)
    assert not eng.hmc_is_warming_up() and eng.hmc_iterations() == 5
    eng.hmc_set_n_leapfrog(0)                    # l.max(1)
    eng.hmc_step(30)                             # crosses iteration 20, where the mass reset was armed
    assert np.array_equal(eng.hmc_mass(), np.ones((cp.d, 64)))
    assert np.array_equal(eng.hmc_step_sizes(), np.full(64, 0.07))
    pos, info = eng.hmc_step_info(2)
    assert np.array_equal(info["step_size"], np.full((2, 64), 0.07))
