"""ctypes wrapper around the CPU ORACLE (oracle/libfugue_oracle.so).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package `fugue_amd`.
It lowers a `fugue_amd.model.Program` *description* onto the oracle's own expression-tree
model (independent of the product's site-program compiler).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FUGUE_ORACLE_LIB: another build of the same sources (the sanitizer build `make asan`, tests/test_sanitizers_cpu.py)
_LIB_PATH = os.environ.get("FUGUE_ORACLE_LIB") or os.path.join(_HERE, "libfugue_oracle.so")

DISTS = ["Bernoulli", "Beta", "Binomial", "Categorical", "Cauchy", "ChiSquared", "DiscreteUniform",
         "Exponential", "Gamma", "InverseGamma", "Laplace", "LogNormal", "Normal", "Poisson",
         "StudentT", "Uniform", "Weibull"]
XOPS = {"const": 0, "site": 1, "data": 2, "neg": 3, "add": 4, "sub": 5, "mul": 6, "div": 7, "exp": 8,
        "ln": 9, "sqrt": 10, "abs": 11, "floor": 12, "sin": 13, "cos": 14, "tanh": 15, "pow": 16,
        "min": 17, "max": 18, "clamp": 19, "select": 20}
INT_DISTS = {"Bernoulli", "Categorical", "Binomial", "Poisson", "DiscreteUniform"}


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("orc_numerics.c", "orc_model.c", "orc_inference.c",
                                              "fugue_oracle.h", "orc_internal.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if (force or stale) and not os.environ.get("FUGUE_ORACLE_LIB"):
        subprocess.run(["make", "-C", _HERE, "-s", "libfugue_oracle.so"], check=True)
    return _LIB_PATH


class HmcConfig(C.Structure):
    _fields_ = [("n_leapfrog", C.c_int32), ("target_accept", C.c_double), ("init_step_size", C.c_double),
                ("finite_diff_eps", C.c_double), ("adapt_mass", C.c_int32)]

    @classmethod
    def default(cls, **kw):
        c = cls(16, 0.8, float("nan"), 1e-5, 0)
        for k, v in kw.items():
            if k == "init_step_size" and v is None:
                v = float("nan")
            setattr(c, k, v)
        return c


class HmcStats(C.Structure):
    _fields_ = [("accept_rate", C.c_double), ("mean_step_size", C.c_double), ("n_divergent", C.c_int64),
                ("n_model_evals", C.c_int64)]


class MhStats(C.Structure):
    _fields_ = [("accept_rate", C.c_double), ("n_model_evals", C.c_int64)]


class SiteProposal(C.Structure):
    _fields_ = [("kind", C.c_int32), ("lower", C.c_double), ("upper", C.c_double)]


class SmcConfig(C.Structure):
    _fields_ = [("resampling_method", C.c_int32), ("ess_threshold", C.c_double),
                ("rejuvenation_steps", C.c_int32), ("batched_adaptation", C.c_int32)]


class Stream(C.Structure):
    _fields_ = [("key0", C.c_uint32), ("key1", C.c_uint32), ("c0", C.c_uint32), ("c1", C.c_uint32),
                ("c2", C.c_uint32), ("c3", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    dp, ip, i64p = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int64)
    vp = C.c_void_p
    L.orc_logpdf.restype = C.c_double
    L.orc_logpdf.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int64, dp, C.c_int]
    L.orc_logpdf_du.restype = C.c_double
    L.orc_logpdf_du.argtypes = [C.c_int64, C.c_int64, C.c_int64]
    L.orc_log_sum_exp.restype = C.c_double
    L.orc_log_sum_exp.argtypes = [dp, C.c_size_t]
    L.orc_normalize_log_probs.argtypes = [dp, C.c_size_t, dp]
    L.orc_log1p_exp.restype = C.c_double
    L.orc_log1p_exp.argtypes = [C.c_double]
    L.orc_safe_ln.restype = C.c_double
    L.orc_safe_ln.argtypes = [C.c_double]
    L.orc_stream_init.argtypes = [C.POINTER(Stream), C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
    L.orc_stream_u01.restype = C.c_double
    L.orc_stream_u01.argtypes = [C.POINTER(Stream)]
    L.orc_stream_normal.restype = C.c_double
    L.orc_stream_normal.argtypes = [C.POINTER(Stream)]
    L.orc_stream_gaussian_z.restype = C.c_double
    L.orc_stream_gaussian_z.argtypes = [C.POINTER(Stream)]
    L.orc_stream_block.argtypes = [C.POINTER(Stream), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.orc_sample_dist.restype = C.c_int64      # orc_cell returned as raw 8 bytes (INTEGER class)
    L.orc_sample_dist.argtypes = [C.c_int, dp, C.c_int, C.POINTER(Stream)]
    L.orc_model_new.restype = vp
    L.orc_model_free.argtypes = [vp]
    L.orc_model_add_data.argtypes = [vp, dp, C.c_int]
    L.orc_model_add_node.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
    L.orc_model_add_args.argtypes = [vp, ip, C.c_int]
    L.orc_model_add_stmt.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, ip, C.c_int, C.c_int]
    L.orc_model_finalize.argtypes = [vp]
    for f in ("orc_model_n_sites", "orc_model_n_f64", "orc_model_n_observe"):
        getattr(L, f).argtypes = [vp]
    L.orc_model_site_name.restype = C.c_char_p
    L.orc_model_site_name.argtypes = [vp, C.c_int]
    L.orc_model_site_vtype.argtypes = [vp, C.c_int]
    L.orc_model_site_of_handle.argtypes = [vp, C.c_int]
    L.orc_model_f64_site.argtypes = [vp, C.c_int]
    L.orc_run_score.argtypes = [vp, vp, dp, dp]
    L.orc_run_prior.argtypes = [vp, C.POINTER(Stream), vp, dp, dp]
    L.orc_log_joint_at.restype = C.c_double
    L.orc_log_joint_at.argtypes = [vp, vp, dp]
    L.orc_grad_log_joint.argtypes = [vp, vp, dp, C.c_double, dp]
    L.orc_leapfrog.argtypes = [vp, vp, dp, dp, C.c_double, C.c_int, C.c_double, dp, dp, dp]
    L.orc_hmc_transition.argtypes = [vp, vp, dp, C.c_double, C.c_double, C.c_int, C.c_double, dp, dp,
                                     C.c_double, dp, dp, ip, dp, ip]
    L.orc_find_reasonable_epsilon.restype = C.c_double
    L.orc_find_reasonable_epsilon.argtypes = [vp, vp, dp, C.c_double, C.c_double, dp, dp]
    L.orc_hmc_momentum.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, dp, C.c_int, dp, dp]
    L.orc_dual_averaging_run.restype = C.c_double
    L.orc_dual_averaging_run.argtypes = [C.c_double, C.c_double, dp, C.c_int, dp, dp]
    L.orc_hmc_run.argtypes = [vp, C.POINTER(HmcConfig), C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int,
                              dp, vp, dp, C.POINTER(HmcStats), C.c_int]
    L.orc_adapt_update.restype = C.c_double
    L.orc_adapt_update.argtypes = [dp, dp, i64p, i64p, C.c_int, C.c_double, C.c_double]
    L.orc_mh_run.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(SiteProposal),
                             ip, C.c_int, vp, vp, dp, C.POINTER(MhStats), C.c_int]
    L.orc_systematic_indices.argtypes = [dp, C.c_int64, C.c_double, i64p]
    L.orc_stratified_indices.argtypes = [dp, C.c_int64, dp, i64p]
    L.orc_multinomial_indices.argtypes = [dp, C.c_int64, dp, i64p]
    L.orc_ess_particles.restype = C.c_double
    L.orc_ess_particles.argtypes = [dp, C.c_int64]
    L.orc_next_beta.restype = C.c_double
    L.orc_next_beta.argtypes = [C.c_double, dp, dp, C.c_int64, C.c_double]
    L.orc_smc_run.argtypes = [vp, C.c_int64, C.POINTER(SmcConfig), C.c_uint64, vp, dp, dp, dp, dp, C.c_int,
                              i64p]
    for f in ("orc_split_rhat", "orc_classic_rhat", "orc_ess_multichain"):
        getattr(L, f).restype = C.c_double
        getattr(L, f).argtypes = [dp, C.c_int, C.c_int]
    for f in ("orc_ess_single", "orc_geweke"):
        getattr(L, f).restype = C.c_double
        getattr(L, f).argtypes = [dp, C.c_int]
    L.orc_summarize.argtypes = [dp, C.c_int, C.c_int, dp]
    _lib = L
    return L


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _d(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


# ------------------------------------------------------------------ scalar numerics
def logpdf(dist: str, x, params: Sequence[float]) -> float:
    p = _d(list(params) if len(params) else [0.0])
    k = DISTS.index(dist)
    if dist in INT_DISTS:
        return lib().orc_logpdf(k, 1, 0.0, int(x), _dp(p), len(params))
    return lib().orc_logpdf(k, 0, float(x), 0, _dp(p), len(params))


def logpdf_discrete_uniform(x: int, lo: int, hi: int) -> float:
    return lib().orc_logpdf_du(int(x), int(lo), int(hi))


def log_sum_exp(xs) -> float:
    a = _d(xs)
    return lib().orc_log_sum_exp(_dp(a), a.size)


def normalize_log_probs(xs) -> np.ndarray:
    a = _d(xs)
    out = np.empty_like(a)
    lib().orc_normalize_log_probs(_dp(a), a.size, _dp(out))
    return out


def log1p_exp(x: float) -> float:
    return lib().orc_log1p_exp(float(x))


def safe_ln(x: float) -> float:
    return lib().orc_safe_ln(float(x))


def stream(seed: int, chain: int, it: int, purpose: int) -> Stream:
    s = Stream()
    lib().orc_stream_init(C.byref(s), seed, chain, it, purpose)
    return s


def philox(ctr, key) -> List[int]:
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)


def sample_dist(dist: str, params, s: Stream):
    p = _d(list(params) if len(params) else [0.0])
    raw = lib().orc_sample_dist(DISTS.index(dist), _dp(p), len(params), C.byref(s))
    if dist in INT_DISTS:
        return int(raw)
    return float(np.array([raw], dtype=np.int64).view(np.float64)[0])


# ------------------------------------------------------------------ model
class OracleModel:
    """Oracle-side model built from a `fugue_amd.model.Program` description."""

    def __init__(self, program):
        L = lib()
        self.L = L
        self.h = L.orc_model_new()
        self.program = program
        for arr in program.data:
            a = _d(arr)
            L.orc_model_add_data(self.h, _dp(a), a.size)
        self._memo = {}
        for st in program.stmts:
            if st.kind == 2:
                val = self._node(st.value)
                L.orc_model_add_stmt(self.h, 2, 0, None, None, 0, val)
                continue
            params = [self._node(p) for p in st.dist.params]
            arr = (C.c_int * max(1, len(params)))(*params)
            val = self._node(st.value) if st.kind == 1 else -1
            L.orc_model_add_stmt(self.h, st.kind, DISTS.index(st.dist.name), st.addr.encode("utf-8"), arr,
                                 len(params), val)
        rc = L.orc_model_finalize(self.h)
        if rc != 0:
            raise ValueError(f"oracle finalize failed: ErrorCode {rc}")
        self.S = L.orc_model_n_sites(self.h)
        self.d = L.orc_model_n_f64(self.h)
        self.O = L.orc_model_n_observe(self.h)
        self.site_names = [L.orc_model_site_name(self.h, j).decode("utf-8") for j in range(self.S)]
        self.site_vtypes = [L.orc_model_site_vtype(self.h, j) for j in range(self.S)]
        self.f64_sites = [L.orc_model_f64_site(self.h, k) for k in range(self.d)]

    def __del__(self):
        try:
            self.L.orc_model_free(self.h)
        except Exception:
            pass

    def _node(self, e) -> int:
        key = id(e)
        if key in self._memo:
            return self._memo[key]
        L, h = self.L, self.h
        if e.op == "const":
            r = L.orc_model_add_node(h, 0, 0, 0, 0, e.value)
        elif e.op == "site":
            r = L.orc_model_add_node(h, 1, e.a, 0, 0, 0.0)
        elif e.op == "data":
            r = L.orc_model_add_node(h, 2, e.a, e.b, 0, 0.0)
        elif e.op == "select":
            ids = [self._node(a) for a in e.args]
            arr = (C.c_int * (len(ids) - 1))(*ids[1:])
            start = L.orc_model_add_args(h, arr, len(ids) - 1)
            r = L.orc_model_add_node(h, XOPS["select"], ids[0], start, len(ids) - 1, 0.0)
        else:
            ids = [self._node(a) for a in e.args] + [0, 0, 0]
            r = L.orc_model_add_node(h, XOPS[e.op], ids[0], ids[1], ids[2], 0.0)
        self._memo[key] = r
        return r

    # cells: int64 view; f64 sites hold the double's bits ------------------------------
    def cells(self, values) -> np.ndarray:
        """values: per sorted site python numbers -> int64[S] raw cells."""
        out = np.zeros(self.S, dtype=np.int64)
        for j, v in enumerate(values):
            if self.site_vtypes[j] == 0:
                out[j] = np.array([v], dtype=np.float64).view(np.int64)[0]
            else:
                out[j] = int(v)
        return out

    def decode(self, cells: np.ndarray) -> list:
        cells = np.asarray(cells, dtype=np.int64)
        return [float(cells[j:j + 1].view(np.float64)[0]) if self.site_vtypes[j] == 0 else int(cells[j])
                for j in range(self.S)]

    def run_score(self, cells: np.ndarray):
        cells = np.ascontiguousarray(cells, dtype=np.int64)
        acc = np.zeros(3)
        logp = np.zeros(max(1, self.S))
        self.L.orc_run_score(self.h, cells.ctypes.data, _dp(acc), _dp(logp))
        return acc, logp[:self.S]

    def run_prior(self, seed: int, chain: int, it: int = 0, purpose: int = 1):
        s = stream(seed, chain, it, purpose)
        cells = np.zeros(max(1, self.S), dtype=np.int64)
        acc = np.zeros(3)
        logp = np.zeros(max(1, self.S))
        self.L.orc_run_prior(self.h, C.byref(s), cells.ctypes.data, _dp(acc), _dp(logp))
        return cells[:self.S], acc, logp[:self.S]

    def log_joint_at(self, base_cells, q) -> float:
        b = np.ascontiguousarray(base_cells, dtype=np.int64)
        q = _d(q)
        return self.L.orc_log_joint_at(self.h, b.ctypes.data, _dp(q))

    def grad_log_joint(self, base_cells, q, h=1e-5):
        b = np.ascontiguousarray(base_cells, dtype=np.int64)
        q = _d(q)
        g = np.zeros(max(1, self.d))
        ok = self.L.orc_grad_log_joint(self.h, b.ctypes.data, _dp(q), h, _dp(g))
        return g[:self.d], bool(ok)

    def leapfrog(self, base_cells, q0, p0, eps, l, h=1e-5, m_inv=None):
        b = np.ascontiguousarray(base_cells, dtype=np.int64)
        q0, p0 = _d(q0), _d(p0)
        mi = _d(m_inv if m_inv is not None else np.ones(self.d))
        q, p = np.zeros(max(1, self.d)), np.zeros(max(1, self.d))
        div = self.L.orc_leapfrog(self.h, b.ctypes.data, _dp(q0), _dp(p0), eps, l, h, _dp(mi), _dp(q), _dp(p))
        return q[:self.d], p[:self.d], bool(div)

    def hmc_transition(self, base_cells, q, lj, eps, l, p0, u, h=1e-5, m_inv=None):
        b = np.ascontiguousarray(base_cells, dtype=np.int64)
        q, p0 = _d(q), _d(p0)
        mi = _d(m_inv if m_inv is not None else np.ones(self.d))
        qo = np.zeros(max(1, self.d))
        ljo, alpha = C.c_double(), C.c_double()
        acc, div = C.c_int(), C.c_int()
        self.L.orc_hmc_transition(self.h, b.ctypes.data, _dp(q), lj, eps, l, h, _dp(mi), _dp(p0), u, _dp(qo),
                                  C.byref(ljo), C.byref(acc), C.byref(alpha), C.byref(div))
        return qo[:self.d], ljo.value, bool(acc.value), alpha.value, bool(div.value)

    def find_reasonable_epsilon(self, base_cells, q, lj, p0, h=1e-5, m_inv=None) -> float:
        b = np.ascontiguousarray(base_cells, dtype=np.int64)
        q, p0 = _d(q), _d(p0)
        mi = _d(m_inv if m_inv is not None else np.ones(self.d))
        return self.L.orc_find_reasonable_epsilon(self.h, b.ctypes.data, _dp(q), lj, h, _dp(mi), _dp(p0))

    def hmc_run(self, seed, n_chains, n_warmup, n_samples, cfg: Optional[HmcConfig] = None, chain0=0,
                n_threads=1, want_draws=True):
        cfg = cfg or HmcConfig.default()
        draws = np.zeros((n_samples, max(1, self.d), n_chains)) if want_draws else None
        final = np.zeros((max(1, self.S), n_chains), dtype=np.int64)
        eps = np.zeros(n_chains)
        st = HmcStats()
        self.L.orc_hmc_run(self.h, C.byref(cfg), seed, chain0, n_chains, n_warmup, n_samples,
                           _dp(draws) if want_draws else None, final.ctypes.data, _dp(eps), C.byref(st),
                           n_threads)
        return (draws[:, :self.d, :] if want_draws else None), final[:self.S], eps, st

    def mh_run(self, seed, n_chains, n_warmup, n_samples, overrides=None, rec_sites=None, chain0=0,
               n_threads=1, want_draws=True):
        rec = list(range(self.S)) if rec_sites is None else list(rec_sites)
        rec_arr = (C.c_int * max(1, len(rec)))(*rec)
        draws = np.zeros((n_samples, max(1, len(rec)), n_chains), dtype=np.int64) if want_draws else None
        final = np.zeros((max(1, self.S), n_chains), dtype=np.int64)
        scales = np.zeros((max(1, self.S), n_chains))
        ov = None
        if overrides is not None:
            ov = (SiteProposal * self.S)()
            for j, o in enumerate(overrides):
                ov[j] = SiteProposal(*o) if o is not None else SiteProposal(0, 0.0, 0.0)
        st = MhStats()
        self.L.orc_mh_run(self.h, seed, chain0, n_chains, n_warmup, n_samples, ov, rec_arr, len(rec),
                          draws.ctypes.data if want_draws else None, final.ctypes.data, _dp(scales),
                          C.byref(st), n_threads)
        return (draws[:, :len(rec), :] if want_draws else None), final[:self.S], scales[:self.S], st

    def smc_run(self, n, seed, method=1, ess_threshold=0.5, rejuvenation_steps=0, batched=0, max_betas=10000):
        cfg = SmcConfig(method, ess_threshold, rejuvenation_steps, batched)
        values = np.zeros((max(1, self.S), n), dtype=np.int64)
        log_w, weights = np.zeros(n), np.zeros(n)
        logz = C.c_double()
        betas = np.zeros(max_betas)
        evals = C.c_int64()
        k = self.L.orc_smc_run(self.h, n, C.byref(cfg), seed, values.ctypes.data, _dp(log_w), _dp(weights),
                               C.byref(logz), _dp(betas), max_betas, C.byref(evals))
        return dict(values=values[:self.S], log_w=log_w, weights=weights, log_evidence=logz.value,
                    betas=betas[:k], n_model_evals=evals.value)


# ------------------------------------------------------------------ resampling / SMC pieces
def systematic_indices(w, U: float) -> np.ndarray:
    w = _d(w)
    idx = np.zeros(w.size, dtype=np.int64)
    lib().orc_systematic_indices(_dp(w), w.size, U, idx.ctypes.data_as(C.POINTER(C.c_int64)))
    return idx


def stratified_indices(w, U) -> np.ndarray:
    w, U = _d(w), _d(U)
    idx = np.zeros(w.size, dtype=np.int64)
    lib().orc_stratified_indices(_dp(w), w.size, _dp(U), idx.ctypes.data_as(C.POINTER(C.c_int64)))
    return idx


def multinomial_indices(w, U) -> np.ndarray:
    w, U = _d(w), _d(U)
    idx = np.zeros(w.size, dtype=np.int64)
    lib().orc_multinomial_indices(_dp(w), w.size, _dp(U), idx.ctypes.data_as(C.POINTER(C.c_int64)))
    return idx


def ess_particles(w) -> float:
    w = _d(w)
    return lib().orc_ess_particles(_dp(w), w.size)


def next_beta(beta, log_w, ll, target_ess) -> float:
    log_w, ll = _d(log_w), _d(ll)
    return lib().orc_next_beta(beta, _dp(log_w), _dp(ll), log_w.size, target_ess)


def adapt_update(scale, log_scale, acc, tot, accepted, target=0.44, gamma=0.7):
    s, ls = C.c_double(scale), C.c_double(log_scale)
    a, t = C.c_int64(acc), C.c_int64(tot)
    lib().orc_adapt_update(C.byref(s), C.byref(ls), C.byref(a), C.byref(t), int(accepted), target, gamma)
    return s.value, ls.value, a.value, t.value


def hmc_momentum(seed, chain, it, d, mass_sqrt=None, purpose=2):
    """(p0[d], u) of HMC transition `it` of chain `chain` (purpose 2) / eps-search instance (purpose 3)."""
    p0 = np.zeros(max(1, d))
    u = C.c_double()
    ms = _d(mass_sqrt) if mass_sqrt is not None else None
    lib().orc_hmc_momentum(seed, chain, it, purpose, _dp(ms) if ms is not None else None, d, _dp(p0), C.byref(u))
    return p0[:d], u.value


def dual_averaging(eps0, target, alphas):
    a = _d(alphas)
    tr = np.zeros(max(1, a.size))
    fr = C.c_double()
    last = lib().orc_dual_averaging_run(eps0, target, _dp(a), a.size, _dp(tr), C.byref(fr))
    return last, tr[:a.size], fr.value


# ------------------------------------------------------------------ diagnostics
def _chains(chains):
    a = _d(chains)
    assert a.ndim == 2
    return a, a.shape[0], a.shape[1]


def split_rhat(chains) -> float:
    a, m, n = _chains(chains)
    return lib().orc_split_rhat(_dp(a), m, n)


def classic_rhat(chains) -> float:
    a, m, n = _chains(chains)
    return lib().orc_classic_rhat(_dp(a), m, n)


def ess_multichain(chains) -> float:
    a, m, n = _chains(chains)
    return lib().orc_ess_multichain(_dp(a), m, n)


def ess_single(x) -> float:
    a = _d(x)
    return lib().orc_ess_single(_dp(a), a.size)


def geweke(x) -> float:
    a = _d(x)
    return lib().orc_geweke(_dp(a), a.size)


def summarize(chains) -> dict:
    a, m, n = _chains(chains)
    out = np.zeros(9)
    lib().orc_summarize(_dp(a), m, n, _dp(out))
    keys = ["mean", "std", "q2.5", "q25", "q50", "q75", "q97.5", "r_hat", "ess"]
    return dict(zip(keys, out.tolist()))
