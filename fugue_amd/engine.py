"""ctypes binding of the C ABI (include/fugue_amd.h) + lowering of a `model.Program`
description onto it.  This is host plumbing only: every density, gradient and transition is
computed by the HIP kernels in fugue_amd/lib/libfugue_amd.so.  There is no CPU fallback --
loading fails loudly when the library is missing, and engine creation fails loudly when no
gfx950 device is present.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

from . import model as M

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FG_LIB_PATH") or os.path.join(_HERE, "lib", "libfugue_amd.so")   # FG_LIB_PATH: an experiment build (tools/)


class fg_tok(C.Structure):
    _fields_ = [("op", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("reserved", C.c_int32), ("imm", C.c_double)]


class fg_hmc_config(C.Structure):
    _fields_ = [("n_leapfrog", C.c_int32), ("target_accept", C.c_double), ("init_step_size", C.c_double),
                ("finite_diff_eps", C.c_double), ("adapt_mass", C.c_int32), ("grad_mode", C.c_int32)]


class fg_hmc_stats(C.Structure):
    _fields_ = [("accept_rate", C.c_double), ("mean_step_size", C.c_double), ("n_divergent", C.c_int64),
                ("n_transitions", C.c_int64)]


class fg_site_proposal(C.Structure):
    _fields_ = [("kind", C.c_int32), ("lower", C.c_double), ("upper", C.c_double)]


class fg_mh_stats(C.Structure):
    _fields_ = [("accept_rate", C.c_double), ("n_steps", C.c_int64)]


class fg_smc_config(C.Structure):
    _fields_ = [("resampling_method", C.c_int32), ("ess_threshold", C.c_double), ("rejuvenation_steps", C.c_int32), ("sequential_adaptation", C.c_int32)]


class fg_smc_result(C.Structure):
    _fields_ = [("log_evidence", C.c_double), ("n_steps", C.c_int32), ("n_model_runs", C.c_int64)]


RESAMPLE_MULTINOMIAL, RESAMPLE_SYSTEMATIC, RESAMPLE_STRATIFIED = range(3)
PROP_AUTO, PROP_GAUSSIAN, PROP_LOGSPACE, PROP_REFLECT, PROP_PRIOR_RESAMPLE = range(5)

TOK = {"const": 0, "site": 1, "data": 2, "neg": 3, "exp": 4, "ln": 5, "sqrt": 6, "abs": 7, "floor": 8, "sin": 9,
       "cos": 10, "tanh": 11, "add": 12, "sub": 13, "mul": 14, "div": 15, "pow": 16, "min": 17, "max": 18,
       "clamp": 19, "select": 20}
GRAD_FD_DENSE, GRAD_FD_SPARSE, GRAD_ANALYTIC = 0, 1, 2
# engine error codes (include/fugue_amd.h)
FG_E_NO_DEVICE, FG_E_HIP, FG_E_BAD_ARG, FG_E_NOT_FINALIZED, FG_E_STATE, FG_E_UNSUPPORTED, FG_E_LIMIT = -1, -2, -3, -4, -5, -6, -7

# every symbol include/fugue_amd.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "fg_program_new", "fg_program_free", "fg_program_data", "fg_program_sample", "fg_program_observe",
    "fg_program_factor", "fg_program_finalize", "fg_program_n_sites", "fg_program_n_f64", "fg_program_n_observe",
    "fg_program_n_instructions", "fg_program_n_slots", "fg_program_site_name", "fg_program_site_vtype",
    "fg_program_site_of_handle", "fg_program_f64_site", "fg_program_dep_count", "fg_program_stream_records", "fg_last_error", "fg_abi_version",
    "fg_engine_new", "fg_engine_free", "fg_engine_synchronize", "fg_engine_stream", "fg_engine_set_stream", "fg_engine_n_chains",
    "fg_engine_set_values", "fg_engine_get_values", "fg_engine_values_device", "fg_prior_init", "fg_log_joint", "fg_log_joint_stream",
    "fg_program_sample_discrete_uniform",
    "fg_hmc_config_default", "fg_hmc_init", "fg_hmc_step", "fg_hmc_step_info", "fg_hmc_get_mass", "fg_hmc_run", "fg_hmc_get_stats", "fg_hmc_get_step_sizes",
    "fg_hmc_get_log_joint", "fg_hmc_set_step_size", "fg_hmc_set_n_leapfrog", "fg_hmc_is_warming_up", "fg_hmc_iterations", "fg_hmc_step_recorded",
    "fg_state_size", "fg_state_export", "fg_state_import", "fg_hmc_grad", "fg_hmc_transition_injected",
    "fg_hmc_find_eps_injected", "fg_mh_init", "fg_mh_step", "fg_mh_set_recording", "fg_mh_run", "fg_mh_get_stats", "fg_mh_get_scales",
    "fg_mh_get_log_weight", "fg_smc_config_default", "fg_smc_run", "fg_smc_prior_particles", "fg_smc_normalize", "fg_smc_ess", "fg_smc_resample", "fg_smc_rejuvenate",
    "fg_smc_get_weights", "fg_smc_set_log_weights", "fg_device_log_sum_exp", "fg_device_next_beta",
    "fg_device_resample_indices", "fg_diag_chain_moments", "fg_diag_autocov_sums", "fg_diag_rhat_ess", "fg_diag_combine", "fg_diag_geweke",
    "fg_diag_combine_reduced", "fg_diag_set_exchange", "fg_diag_exchange_bytes", "fg_hmc_last_kernel", "fg_mh_last_kernel", "fg_diag_quantiles",
    "fg_comm_unique_id", "fg_comm_init", "fg_comm_destroy", "fg_device_alloc", "fg_device_free", "fg_device_download", "fg_device_upload",
    "fg_dsl_compile", "fg_dsl_warning_count", "fg_dsl_warning",
]

_lib = None
ACOV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double))     # fg_acov_fn
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double))     # fg_reduce_fn
DIAG_REDUCE, DIAG_GATHER = 0, 1


class DslError(ValueError):
    """Static error of the model DSL (syntax, unknown name, arity); message as the reference words it."""


class EngineError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"fugue_amd error {code}: {message}")
        self.code = code


def lib():
    """Loads libfugue_amd.so (building it in-tree first if hipcc is available and it is stale)."""
    global _lib
    if _lib is not None:
        return _lib
    from . import build as _b
    try:                                  # a no-op when the library is fresh; rebuilds a stale one where hipcc exists
        if not (os.environ.get("FG_LIB_PATH") and os.path.exists(LIB_PATH)):   # an experiment library is taken as it is
            _b.build()
    except Exception:
        if not os.path.exists(LIB_PATH):
            raise
    L = C.CDLL(LIB_PATH)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)
    tp = C.POINTER(fg_tok)
    L.fg_last_error.restype = C.c_char_p
    L.fg_program_new.restype = vp
    L.fg_program_free.argtypes = [vp]
    L.fg_program_data.argtypes = [vp, C.c_char_p, dp, C.c_int64]
    L.fg_program_sample.argtypes = [vp, C.c_char_p, C.c_int, tp, ip, C.c_int]
    L.fg_program_observe.argtypes = [vp, C.c_char_p, C.c_int, tp, ip, C.c_int, tp, C.c_int]
    L.fg_program_factor.argtypes = [vp, tp, C.c_int]
    L.fg_program_sample_discrete_uniform.argtypes = [vp, C.c_char_p, C.c_int64, C.c_int64]
    for f in ("fg_program_finalize", "fg_program_n_sites", "fg_program_n_f64", "fg_program_n_observe",
              "fg_program_n_instructions", "fg_program_n_slots"):
        getattr(L, f).argtypes = [vp]
    L.fg_program_site_name.argtypes = [vp, C.c_int, C.c_char_p, C.c_int]
    for f in ("fg_program_site_vtype", "fg_program_site_of_handle", "fg_program_f64_site", "fg_program_dep_count", "fg_program_stream_records"):
        getattr(L, f).argtypes = [vp, C.c_int]
    L.fg_engine_new.restype = vp
    L.fg_engine_new.argtypes = [vp, C.c_int64, C.c_uint64, C.c_uint32, C.c_int]
    L.fg_engine_free.argtypes = [vp]
    L.fg_engine_synchronize.argtypes = [vp]
    L.fg_engine_stream.restype = vp
    L.fg_engine_stream.argtypes = [vp]
    L.fg_engine_set_stream.argtypes = [vp, vp]
    L.fg_engine_n_chains.restype = C.c_int64
    L.fg_engine_n_chains.argtypes = [vp]
    L.fg_engine_set_values.argtypes = [vp, vp]
    L.fg_engine_get_values.argtypes = [vp, vp]
    L.fg_engine_values_device.restype = vp
    L.fg_engine_values_device.argtypes = [vp]
    L.fg_prior_init.argtypes = [vp, C.c_uint32, dp]
    L.fg_log_joint.argtypes = [vp, dp, dp]
    L.fg_log_joint_stream.argtypes = [vp, dp, dp]
    L.fg_hmc_config_default.argtypes = [C.POINTER(fg_hmc_config)]
    L.fg_hmc_init.argtypes = [vp, C.POINTER(fg_hmc_config), C.c_int]
    L.fg_hmc_step.argtypes = [vp, C.c_int, vp]
    L.fg_hmc_step_info.argtypes = [vp, C.c_int, vp, vp]
    L.fg_hmc_get_mass.argtypes = [vp, dp]
    L.fg_hmc_run.argtypes = [vp, C.POINTER(fg_hmc_config), C.c_int, C.c_int, vp, C.POINTER(fg_hmc_stats)]
    L.fg_hmc_get_stats.argtypes = [vp, C.POINTER(fg_hmc_stats)]
    L.fg_hmc_get_step_sizes.argtypes = [vp, dp]
    L.fg_hmc_get_log_joint.argtypes = [vp, dp]
    L.fg_hmc_set_step_size.argtypes = [vp, C.c_double]
    L.fg_hmc_set_n_leapfrog.argtypes = [vp, C.c_int]
    L.fg_hmc_is_warming_up.argtypes = [vp]
    L.fg_hmc_iterations.restype = C.c_int64
    L.fg_hmc_iterations.argtypes = [vp]
    L.fg_hmc_step_recorded.argtypes = [vp, C.c_int, C.POINTER(C.c_int64), dp, dp, ip, vp]
    L.fg_state_size.restype = C.c_int64
    L.fg_state_size.argtypes = [vp]
    L.fg_state_export.argtypes = [vp, vp, C.c_size_t]
    L.fg_state_import.argtypes = [vp, vp, C.c_size_t]
    L.fg_hmc_grad.argtypes = [vp, C.c_double, C.c_int, dp, ip]
    L.fg_hmc_transition_injected.argtypes = [vp, C.POINTER(fg_hmc_config), C.c_double, dp, dp, ip, dp, ip, dp]
    L.fg_hmc_find_eps_injected.argtypes = [vp, C.POINTER(fg_hmc_config), dp, dp]
    L.fg_mh_init.argtypes = [vp, C.c_int, C.POINTER(fg_site_proposal)]
    L.fg_mh_step.argtypes = [vp, C.c_int, ip, C.c_int, vp]
    L.fg_mh_set_recording.argtypes = [vp, C.c_int]
    L.fg_mh_run.argtypes = [vp, C.c_int, C.c_int, C.POINTER(fg_site_proposal), ip, C.c_int, vp, C.POINTER(fg_mh_stats)]
    L.fg_mh_get_stats.argtypes = [vp, C.POINTER(fg_mh_stats)]
    L.fg_mh_get_scales.argtypes = [vp, dp]
    L.fg_mh_get_log_weight.argtypes = [vp, dp]
    L.fg_smc_config_default.argtypes = [C.POINTER(fg_smc_config)]
    L.fg_smc_run.argtypes = [vp, C.POINTER(fg_smc_config), dp, dp, C.POINTER(fg_smc_result), dp, C.c_int]
    L.fg_smc_prior_particles.argtypes = [vp, C.c_uint32]
    L.fg_smc_normalize.argtypes = [vp]
    L.fg_smc_ess.argtypes = [vp, dp]
    L.fg_smc_resample.argtypes = [vp, C.c_int, C.c_uint32, C.POINTER(C.c_int64)]
    L.fg_smc_rejuvenate.argtypes = [vp, C.c_double, C.c_int, C.c_uint32, dp]
    L.fg_smc_get_weights.argtypes = [vp, dp, dp]
    L.fg_smc_set_log_weights.argtypes = [vp, dp]
    L.fg_device_log_sum_exp.argtypes = [C.c_int, dp, C.c_int64, dp]
    L.fg_device_next_beta.argtypes = [C.c_int, C.c_double, dp, dp, C.c_int64, C.c_double, dp]
    L.fg_device_resample_indices.argtypes = [C.c_int, C.c_int, dp, C.c_int64, dp, C.POINTER(C.c_int64)]
    L.fg_diag_chain_moments.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    L.fg_diag_autocov_sums.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, dp]
    L.fg_diag_geweke.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    L.fg_diag_rhat_ess.argtypes = [vp, vp, C.c_int, C.c_int, vp, dp, dp, dp, dp, C.POINTER(C.c_int64)]
    L.fg_diag_combine.argtypes = [dp, C.c_int64, C.c_int, C.c_int, ACOV_FN, vp, dp, dp, dp, dp]
    L.fg_diag_combine_reduced.argtypes = [C.c_int64, C.c_int, C.c_int, REDUCE_FN, ACOV_FN, vp, dp, dp, dp, dp]
    L.fg_diag_quantiles.argtypes = [vp, vp, C.c_int, C.c_int, vp, dp, C.c_int, dp]
    L.fg_diag_set_exchange.argtypes = [vp, C.c_int]
    L.fg_diag_exchange_bytes.restype = C.c_int64
    L.fg_diag_exchange_bytes.argtypes = [vp]
    L.fg_hmc_last_kernel.restype = C.c_char_p
    L.fg_hmc_last_kernel.argtypes = [vp]
    L.fg_mh_last_kernel.restype = C.c_char_p
    L.fg_mh_last_kernel.argtypes = [vp]
    L.fg_comm_unique_id.argtypes = [vp]
    L.fg_comm_init.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.fg_comm_destroy.argtypes = [vp]
    L.fg_device_alloc.restype = vp
    L.fg_device_alloc.argtypes = [vp, C.c_size_t]
    L.fg_device_free.argtypes = [vp, vp]
    L.fg_device_download.argtypes = [vp, vp, vp, C.c_size_t]
    L.fg_device_upload.argtypes = [vp, vp, vp, C.c_size_t]
    L.fg_dsl_compile.restype = vp
    L.fg_dsl_compile.argtypes = [C.c_char_p, C.c_char_p]
    L.fg_dsl_warning_count.argtypes = [vp]
    L.fg_dsl_warning.restype = C.c_char_p
    L.fg_dsl_warning.argtypes = [vp, C.c_int]
    _lib = L
    return L


def last_error() -> str:
    return lib().fg_last_error().decode("utf-8", "replace")


def _check(rc: int):
    if rc != 0:
        raise EngineError(rc, last_error())


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def hmc_config(n_leapfrog=16, target_accept=0.8, init_step_size=None, finite_diff_eps=1e-5, adapt_mass=False,
               grad_mode=GRAD_FD_SPARSE) -> fg_hmc_config:
    """`HMCConfig` (/root/reference/src/inference/hmc.rs:106-135), same defaults; grad_mode: the engine's one default is
    FG_GRAD_FD_SPARSE (C ABI, Python, bench) -- GRAD_FD_DENSE is the reference's arithmetic verbatim."""
    return fg_hmc_config(int(n_leapfrog), float(target_accept),
                         float("nan") if init_step_size is None else float(init_step_size),
                         float(finite_diff_eps), int(bool(adapt_mass)), int(grad_mode))


# --------------------------------------------------------------------------------------
def _postfix(e: M.Expr, out: List[fg_tok]):
    if e.op == "const":
        out.append(fg_tok(TOK["const"], 0, 0, 0, e.value))
    elif e.op == "site":
        out.append(fg_tok(TOK["site"], e.a, 0, 0, 0.0))
    elif e.op == "data":
        out.append(fg_tok(TOK["data"], e.a, e.b, 0, 0.0))
    elif e.op == "select":
        for a in e.args:
            _postfix(a, out)
        out.append(fg_tok(TOK["select"], len(e.args) - 1, 0, 0, 0.0))
    else:
        for a in e.args:
            _postfix(a, out)
        out.append(fg_tok(TOK[e.op], 0, 0, 0, 0.0))


def _tok_array(toks: List[fg_tok]):
    arr = (fg_tok * max(1, len(toks)))()
    for i, t in enumerate(toks):
        arr[i] = t
    return arr


class CompiledProgram:
    """A finalized `fg_program` (site program) built from a `model.Program` description."""

    def __init__(self, program: Optional[M.Program], _handle=None):
        L = lib()
        self.program = program
        self.warnings: List[str] = []
        if _handle is not None:           # already finalized by a native front-end (fg_dsl_compile)
            self.h = _handle
            self._describe()
            return
        self.h = L.fg_program_new()
        for name, arr in zip(program.data_names, program.data):
            a = np.ascontiguousarray(arr, dtype=np.float64)
            rc = L.fg_program_data(self.h, name.encode(), _dp(a), a.size)
            if rc < 0:
                raise EngineError(rc, last_error())
        for st in program.stmts:
            if st.kind == M.FACTOR:
                toks: List[fg_tok] = []
                _postfix(st.value, toks)
                _check(L.fg_program_factor(self.h, _tok_array(toks), len(toks)))
                continue
            toks, lens = [], []
            for p in st.dist.params:
                n0 = len(toks)
                _postfix(p, toks)
                lens.append(len(toks) - n0)
            plen = (C.c_int32 * max(1, len(lens)))(*lens)
            if st.kind == M.SAMPLE and st.dist.i64_bounds is not None:
                rc = L.fg_program_sample_discrete_uniform(self.h, st.addr.encode("utf-8"), *st.dist.i64_bounds)
                if rc < 0 or rc != st.handle:
                    raise EngineError(rc, last_error())
            elif st.kind == M.SAMPLE:
                rc = L.fg_program_sample(self.h, st.addr.encode("utf-8"), st.dist.kind, _tok_array(toks), plen, len(lens))
                if rc < 0 or rc != st.handle:
                    raise EngineError(rc, last_error())
            else:
                vt: List[fg_tok] = []
                _postfix(st.value, vt)
                _check(L.fg_program_observe(self.h, st.addr.encode("utf-8"), st.dist.kind, _tok_array(toks), plen,
                                            len(lens), _tok_array(vt), len(vt)))
        rc = L.fg_program_finalize(self.h)
        if rc != 0:
            msg = last_error()
            L.fg_program_free(self.h)
            self.h = None
            raise M.FugueError(msg, rc) if rc > 0 else EngineError(rc, msg)
        self._describe()

    @classmethod
    def from_dsl(cls, source: str, data=None) -> "CompiledProgram":
        """`CompiledModel::compile(source, data_json)` of the playground DSL (crates/fugue-wasm/src/dsl.rs:1062-1120):
        `data` is a JSON string, a dict of arrays, a bare list (bound to `data`) or None."""
        import json
        L = lib()
        dj = data if isinstance(data, str) else ("" if data is None else json.dumps(data))
        h = L.fg_dsl_compile(source.encode("utf-8"), dj.encode("utf-8"))
        if not h:
            raise DslError(last_error())
        cp = cls(None, _handle=h)
        cp.warnings = [L.fg_dsl_warning(h, i).decode("utf-8") for i in range(L.fg_dsl_warning_count(h))]
        return cp

    def _describe(self):
        L = lib()
        self.S = L.fg_program_n_sites(self.h)
        self.d = L.fg_program_n_f64(self.h)
        self.O = L.fg_program_n_observe(self.h)
        self.n_instructions = L.fg_program_n_instructions(self.h)
        self.n_slots = L.fg_program_n_slots(self.h)
        buf = C.create_string_buffer(4096)
        self.site_names = []
        for j in range(self.S):
            L.fg_program_site_name(self.h, j, buf, 4096)
            self.site_names.append(buf.value.decode("utf-8"))
        self.site_vtypes = [L.fg_program_site_vtype(self.h, j) for j in range(self.S)]
        self.f64_sites = [L.fg_program_f64_site(self.h, k) for k in range(self.d)]
        self.dep_counts = [L.fg_program_dep_count(self.h, k) for k in range(self.d)]
        self.stream_records = tuple(L.fg_program_stream_records(self.h, w) for w in range(3))   # (gradient, score, kinds)

    def site_of_handle(self, h: int) -> int:
        return lib().fg_program_site_of_handle(self.h, h)

    def __del__(self):
        try:
            if self.h:
                lib().fg_program_free(self.h)
        except Exception:
            pass

    # cells helpers (same convention as the oracle wrapper: int64 raw view) ---------------
    def cells(self, values: Sequence) -> np.ndarray:
        out = np.zeros(self.S, dtype=np.int64)
        for j, v in enumerate(values):
            out[j] = np.array([v], dtype=np.float64).view(np.int64)[0] if self.site_vtypes[j] == 0 else int(v)
        return out


def compile_model(model_or_fn) -> CompiledProgram:
    prog = model_or_fn if isinstance(model_or_fn, M.Program) else M.trace_model(model_or_fn)
    return CompiledProgram(prog)


class Engine:
    """`fg_engine`: n_chains chains of one program on one MI355X."""

    def __init__(self, compiled: CompiledProgram, n_chains: int, seed: int, chain_offset: int = 0, device: int = 0):
        L = lib()
        self.cp = compiled
        self.C = int(n_chains)
        self.h = L.fg_engine_new(compiled.h, self.C, int(seed) & 0xFFFFFFFFFFFFFFFF, int(chain_offset), int(device))
        if not self.h:
            raise EngineError(-1, last_error())

    def close(self):
        if getattr(self, "h", None):
            lib().fg_engine_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def S(self): return self.cp.S
    @property
    def d(self): return self.cp.d

    def set_stream(self, hip_stream: int):
        _check(lib().fg_engine_set_stream(self.h, hip_stream))

    def synchronize(self):
        _check(lib().fg_engine_synchronize(self.h))

    def set_values(self, cells: np.ndarray):
        a = np.ascontiguousarray(cells, dtype=np.int64)
        assert a.shape == (self.S, self.C), a.shape
        _check(lib().fg_engine_set_values(self.h, a.ctypes.data))

    def get_values(self) -> np.ndarray:
        a = np.zeros((max(1, self.S), self.C), dtype=np.int64)
        _check(lib().fg_engine_get_values(self.h, a.ctypes.data))
        return a[:self.S]

    def prior_init(self, iteration: int = 0) -> np.ndarray:
        acc = np.zeros((3, self.C))
        _check(lib().fg_prior_init(self.h, iteration, _dp(acc)))
        return acc

    def log_joint(self, want_logp: bool = False):
        acc = np.zeros((3, self.C))
        logp = np.zeros((max(1, self.S), self.C)) if want_logp else None
        _check(lib().fg_log_joint(self.h, _dp(acc), _dp(logp) if want_logp else None))
        return (acc, logp[:self.S]) if want_logp else acc

    def log_joint_stream(self, want_records: bool = False):
        """ScoreGivenTrace over the score stream: acc [3][C] (and every statement's log-density [n_records][C])."""
        acc = np.zeros((3, self.C))
        rec = np.zeros((max(1, self.cp.stream_records[1]), self.C)) if want_records else None
        _check(lib().fg_log_joint_stream(self.h, _dp(acc), _dp(rec) if want_records else None))
        return (acc, rec[:self.cp.stream_records[1]]) if want_records else acc

    # ---- HMC ----------------------------------------------------------------------------
    def hmc_init(self, cfg: fg_hmc_config, n_warmup: int):
        _check(lib().fg_hmc_init(self.h, C.byref(cfg), int(n_warmup)))

    def hmc_step(self, n: int, d_draws: Optional[int] = None):
        _check(lib().fg_hmc_step(self.h, int(n), d_draws))

    def hmc_step_info(self, n: int):
        """n transitions with HmcStepInfo: returns positions [n][d][C] and a dict of [n][C] arrays."""
        n = int(n)
        d_pos = self.device_alloc(max(1, n * self.d * self.C) * 8)
        d_info = self.device_alloc(max(1, n * 4 * self.C) * 8)
        try:
            _check(lib().fg_hmc_step_info(self.h, n, d_pos, d_info))
            pos = self.download(d_pos, (n, self.d, self.C))
            info = self.download(d_info, (n, 4, self.C))
        finally:
            self.device_free(d_pos)
            self.device_free(d_info)
        return pos, dict(accepted=info[:, 0] != 0, divergent=info[:, 1] != 0, accept_prob=info[:, 2], step_size=info[:, 3])

    def hmc_mass(self) -> np.ndarray:
        a = np.zeros((max(1, self.d), self.C))
        _check(lib().fg_hmc_get_mass(self.h, _dp(a)))
        return a[:self.d]

    def hmc_run(self, cfg: fg_hmc_config, n_samples: int, n_warmup: int, d_draws: Optional[int] = None) -> fg_hmc_stats:
        st = fg_hmc_stats()
        _check(lib().fg_hmc_run(self.h, C.byref(cfg), int(n_samples), int(n_warmup), d_draws, C.byref(st)))
        return st

    def hmc_stats(self) -> fg_hmc_stats:
        st = fg_hmc_stats()
        _check(lib().fg_hmc_get_stats(self.h, C.byref(st)))
        return st

    def hmc_step_sizes(self) -> np.ndarray:
        a = np.zeros(self.C)
        _check(lib().fg_hmc_get_step_sizes(self.h, _dp(a)))
        return a

    def hmc_log_joint(self) -> np.ndarray:
        a = np.zeros(self.C)
        _check(lib().fg_hmc_get_log_joint(self.h, _dp(a)))
        return a

    def hmc_set_step_size(self, eps: float):
        _check(lib().fg_hmc_set_step_size(self.h, float(eps)))

    def hmc_set_n_leapfrog(self, n: int):
        _check(lib().fg_hmc_set_n_leapfrog(self.h, int(n)))

    def hmc_is_warming_up(self) -> bool:
        return bool(lib().fg_hmc_is_warming_up(self.h))

    def hmc_iterations(self) -> int:
        return int(lib().fg_hmc_iterations(self.h))

    def hmc_step_recorded(self, chain_ids: Sequence[int], n_leapfrog: int):
        """`HmcSession::step_recorded` for every chain; returns (trajectories [K][L+1][d], Hamiltonians [K][L+1],
        points recorded [K]) of the chosen chains."""
        ids = np.ascontiguousarray(chain_ids, dtype=np.int64)
        K, L = ids.size, int(n_leapfrog)
        traj, ham = np.zeros((max(1, K), L + 1, max(1, self.d))), np.zeros((max(1, K), L + 1))
        npts = np.zeros(max(1, K), dtype=np.int32)
        _check(lib().fg_hmc_step_recorded(self.h, K, ids.ctypes.data_as(C.POINTER(C.c_int64)), _dp(traj), _dp(ham),
                                          npts.ctypes.data_as(C.POINTER(C.c_int32)), None))
        return traj[:K, :, :self.d], ham[:K], npts[:K]

    def state_export(self) -> bytes:
        n = lib().fg_state_size(self.h)
        buf = C.create_string_buffer(n)
        _check(lib().fg_state_export(self.h, buf, n))
        return buf.raw

    def state_import(self, blob: bytes):
        _check(lib().fg_state_import(self.h, blob, len(blob)))

    def hmc_grad(self, h: float = 1e-5, grad_mode: int = GRAD_FD_DENSE):
        g = np.zeros((max(1, self.d), self.C))
        ok = np.zeros(self.C, dtype=np.int32)
        _check(lib().fg_hmc_grad(self.h, float(h), int(grad_mode), _dp(g), ok.ctypes.data_as(C.POINTER(C.c_int32))))
        return g[:self.d], ok.astype(bool)

    def hmc_transition_injected(self, cfg: fg_hmc_config, eps: float, p0: np.ndarray, u: np.ndarray):
        p0 = np.ascontiguousarray(p0, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert p0.shape == (self.d, self.C) and u.shape == (self.C,)
        acc = np.zeros(self.C, dtype=np.int32)
        div = np.zeros(self.C, dtype=np.int32)
        alpha, lj = np.zeros(self.C), np.zeros(self.C)
        ip = C.POINTER(C.c_int32)
        _check(lib().fg_hmc_transition_injected(self.h, C.byref(cfg), float(eps), _dp(p0), _dp(u), acc.ctypes.data_as(ip),
                                                _dp(alpha), div.ctypes.data_as(ip), _dp(lj)))
        return acc.astype(bool), alpha, div.astype(bool), lj

    def hmc_find_eps_injected(self, cfg: fg_hmc_config, p0: np.ndarray) -> np.ndarray:
        p0 = np.ascontiguousarray(p0, dtype=np.float64)
        eps = np.zeros(self.C)
        _check(lib().fg_hmc_find_eps_injected(self.h, C.byref(cfg), _dp(p0), _dp(eps)))
        return eps

    # ---- MH -----------------------------------------------------------------------------
    def _overrides(self, overrides):
        if overrides is None:
            return None
        arr = (fg_site_proposal * max(1, self.S))()
        for j, o in enumerate(overrides):
            arr[j] = fg_site_proposal(*o) if o is not None else fg_site_proposal(0, 0.0, 0.0)
        return arr

    def mh_init(self, n_warmup: int, overrides=None):
        _check(lib().fg_mh_init(self.h, int(n_warmup), self._overrides(overrides)))

    def mh_step(self, n: int, rec_sites: Sequence[int] = (), d_draws: Optional[int] = None):
        rec = (C.c_int32 * max(1, len(rec_sites)))(*rec_sites)
        _check(lib().fg_mh_step(self.h, int(n), rec, len(rec_sites), d_draws))

    def mh_set_recording(self, during_adaptation: bool):
        """Record after every step, adapting or not (incremental sessions: fugue_amd.session)."""
        _check(lib().fg_mh_set_recording(self.h, 1 if during_adaptation else 0))

    def mh_run(self, n_samples: int, n_warmup: int, overrides=None, rec_sites: Sequence[int] = (), d_draws=None) -> fg_mh_stats:
        rec = (C.c_int32 * max(1, len(rec_sites)))(*rec_sites)
        st = fg_mh_stats()
        _check(lib().fg_mh_run(self.h, int(n_samples), int(n_warmup), self._overrides(overrides), rec, len(rec_sites),
                               d_draws, C.byref(st)))
        return st

    def mh_stats(self) -> fg_mh_stats:
        st = fg_mh_stats()
        _check(lib().fg_mh_get_stats(self.h, C.byref(st)))
        return st

    def mh_scales(self) -> np.ndarray:
        a = np.zeros((max(1, self.S), self.C))
        _check(lib().fg_mh_get_scales(self.h, _dp(a)))
        return a[:self.S]

    def mh_log_weight(self) -> np.ndarray:
        a = np.zeros(self.C)
        _check(lib().fg_mh_get_log_weight(self.h, _dp(a)))
        return a

    # ---- SMC ----------------------------------------------------------------------------
    def smc_run(self, resampling_method=RESAMPLE_SYSTEMATIC, ess_threshold=0.5, rejuvenation_steps=0, max_betas=10000, download=True, sequential_adaptation=False):
        """`adaptive_smc` (/root/reference/src/inference/smc.rs:455-581) over this engine's chains as particles.
        download=False leaves particles and weights in HBM (engine values / fg_smc_run's device state): only the
        evidence, the ladder and the counters come back."""
        cfg = fg_smc_config(int(resampling_method), float(ess_threshold), int(rejuvenation_steps), 1 if sequential_adaptation else 0)
        res = fg_smc_result()
        betas = np.zeros(max_betas)
        log_w, w = (np.zeros(self.C), np.zeros(self.C)) if download else (None, None)
        _check(lib().fg_smc_run(self.h, C.byref(cfg), _dp(log_w) if download else None, _dp(w) if download else None, C.byref(res), _dp(betas), max_betas))
        return dict(values=self.get_values() if download else None, log_w=log_w, weights=w, log_evidence=res.log_evidence,
                    betas=betas[:res.n_steps], n_model_runs=res.n_model_runs)

    # standalone building blocks (smc.rs:230-349, 698-790) over the engine's particle population
    def smc_prior_particles(self, iteration: int = 0):
        _check(lib().fg_smc_prior_particles(self.h, int(iteration)))

    def smc_normalize(self):
        _check(lib().fg_smc_normalize(self.h))

    def smc_ess(self) -> float:
        out = C.c_double()
        _check(lib().fg_smc_ess(self.h, C.byref(out)))
        return out.value

    def smc_resample(self, method: int = RESAMPLE_SYSTEMATIC, step: int = 1) -> np.ndarray:
        idx = np.zeros(self.C, dtype=np.int64)
        _check(lib().fg_smc_resample(self.h, int(method), int(step), idx.ctypes.data_as(C.POINTER(C.c_int64))))
        return idx

    def smc_rejuvenate(self, beta: float, steps: int, first_move_id: int = 0) -> float:
        out = C.c_double()
        _check(lib().fg_smc_rejuvenate(self.h, float(beta), int(steps), int(first_move_id), C.byref(out)))
        return out.value

    def smc_weights(self):
        lw, w = np.zeros(self.C), np.zeros(self.C)
        _check(lib().fg_smc_get_weights(self.h, _dp(lw), _dp(w)))
        return lw, w

    def smc_set_log_weights(self, log_w):
        a = np.ascontiguousarray(log_w, dtype=np.float64)
        assert a.shape == (self.C,)
        _check(lib().fg_smc_set_log_weights(self.h, _dp(a)))

    # ---- diagnostics kernels -----------------------------------------------------------------
    def diag_chain_moments(self, d_draws: int, n: int, d: int, d_moments: int):
        _check(lib().fg_diag_chain_moments(self.h, d_draws, int(n), int(d), d_moments))

    def diag_autocov_sums(self, d_draws: int, n: int, d: int, d_moments: int, lag0: int, n_lags: int) -> np.ndarray:
        out = np.zeros((d, n_lags))
        _check(lib().fg_diag_autocov_sums(self.h, d_draws, int(n), int(d), d_moments, int(lag0), int(n_lags), _dp(out)))
        return out

    def diag_geweke(self, d_draws: int, n: int, d: int) -> np.ndarray:
        """`geweke_diagnostic` of every (coordinate, chain) on the device: [d][C]."""
        d_z = self.device_alloc(max(1, d * self.C) * 8)
        try:
            _check(lib().fg_diag_geweke(self.h, d_draws, int(n), int(d), d_z))
            return self.download(d_z, (d, self.C))
        finally:
            self.device_free(d_z)

    def diag_rhat_ess(self, d_draws: int, n: int, d: int, comm: Optional[int] = None, exchange: Optional[int] = None):
        """Split R-hat, multi-chain ESS, pooled mean / std of d_draws [n][d][C] over the chains of every rank of `comm`
        (an RCCL communicator from `comm_init`; None = this engine's chains): computed inside the library.  `exchange`:
        DIAG_REDUCE (default: all-reduces of O(d) chain sums) or DIAG_GATHER (all-gather of every chain's moments)."""
        rhat, ess, mean, std = (np.zeros(d) for _ in range(4))
        tot = C.c_int64()
        if exchange is not None:
            _check(lib().fg_diag_set_exchange(self.h, int(exchange)))
        _check(lib().fg_diag_rhat_ess(self.h, d_draws, int(n), int(d), comm, _dp(rhat), _dp(ess), _dp(mean), _dp(std), C.byref(tot)))
        return dict(r_hat=rhat, ess=ess, mean=mean, std=std, chains=tot.value, exchange_bytes=int(lib().fg_diag_exchange_bytes(self.h)))

    def diag_quantiles(self, d_draws: int, n: int, d: int, probs=(0.025, 0.25, 0.5, 0.75, 0.975), comm: Optional[int] = None) -> np.ndarray:
        """summarize_f64_parameter's quantiles (diagnostics.rs:355-371) of every coordinate of d_draws [n][d][C], selected on the
        device over the draws of every rank of `comm`: [d][len(probs)]."""
        pr = np.ascontiguousarray(probs, dtype=np.float64)
        out = np.zeros((d, len(pr)))
        _check(lib().fg_diag_quantiles(self.h, d_draws, int(n), int(d), comm, _dp(pr), len(pr), _dp(out)))
        return out

    def hmc_last_kernel(self) -> str:
        """Kernel (and waves per tile) the engine's last HMC launch ran."""
        return (lib().fg_hmc_last_kernel(self.h) or b"").decode()

    def mh_last_kernel(self) -> str:
        """Kernel the engine's last MH launch ran."""
        return (lib().fg_mh_last_kernel(self.h) or b"").decode()

    def comm_init(self, world_size: int, rank: int, unique_id: bytes) -> int:
        out = C.c_void_p()
        _check(lib().fg_comm_init(self.h, int(world_size), int(rank), unique_id, C.byref(out)))
        return out.value

    # ---- raw device buffers ---------------------------------------------------------------
    def device_alloc(self, nbytes: int) -> int:
        p = lib().fg_device_alloc(self.h, int(nbytes))
        if not p:
            raise EngineError(-2, last_error())
        return p

    def device_free(self, ptr: int):
        _check(lib().fg_device_free(self.h, ptr))

    def download(self, ptr: int, shape, dtype=np.float64) -> np.ndarray:
        out = np.zeros(shape, dtype=dtype)
        _check(lib().fg_device_download(self.h, out.ctypes.data, ptr, out.nbytes))
        return out


# ---- population-wide device primitives (no program needed) -----------------------------------
def device_log_sum_exp(x, device: int = 0) -> float:
    a = np.ascontiguousarray(x, dtype=np.float64)
    out = C.c_double()
    _check(lib().fg_device_log_sum_exp(device, _dp(a), a.size, C.byref(out)))
    return out.value


def device_next_beta(beta, log_w, ll, target_ess, device: int = 0) -> float:
    lw = np.ascontiguousarray(log_w, dtype=np.float64)
    l2 = np.ascontiguousarray(ll, dtype=np.float64)
    out = C.c_double()
    _check(lib().fg_device_next_beta(device, float(beta), _dp(lw), _dp(l2), lw.size, float(target_ess), C.byref(out)))
    return out.value


def device_resample_indices(method: int, weights, u, device: int = 0) -> np.ndarray:
    w = np.ascontiguousarray(weights, dtype=np.float64)
    uu = np.ascontiguousarray(np.atleast_1d(u), dtype=np.float64)
    idx = np.zeros(w.size, dtype=np.int64)
    _check(lib().fg_device_resample_indices(device, int(method), _dp(w), w.size, _dp(uu), idx.ctypes.data_as(C.POINTER(C.c_int64))))
    return idx


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0 calls it; the host distributes the 128 bytes)."""
    buf = C.create_string_buffer(128)
    _check(lib().fg_comm_unique_id(buf))
    return buf.raw


def comm_destroy(comm: int):
    _check(lib().fg_comm_destroy(comm))


def diag_combine(moments: np.ndarray, n: int, acov_sums):
    """`fg_diag_combine`: the C++ R-hat / ESS combination over host moments [d][6][m] of all chains; `acov_sums(lag0, n_lags)`
    returns the pooled lag sums [d][n_lags] (e.g. all-reduced over gloo)."""
    mom = np.ascontiguousarray(moments, dtype=np.float64)
    d, _, m = mom.shape
    err = []

    def cb(_user, lag0, n_lags, out):
        try:
            a = np.ascontiguousarray(acov_sums(int(lag0), int(n_lags)), dtype=np.float64)
            C.memmove(out, a.ctypes.data, d * n_lags * 8)
            return 0
        except Exception as ex:      # pragma: no cover
            err.append(ex)
            return FG_E_BAD_ARG
    fn = ACOV_FN(cb)
    rhat, ess, mean, std = (np.zeros(d) for _ in range(4))
    rc = lib().fg_diag_combine(_dp(mom), m, int(n), d, fn, None, _dp(rhat), _dp(ess), _dp(mean), _dp(std))
    if err:
        raise err[0]
    _check(rc)
    return dict(r_hat=rhat, ess=ess, mean=mean, std=std)


def diag_combine_reduced(m: int, n: int, d: int, reduce, acov_sums):
    """`fg_diag_combine_reduced`: the same combination from sums over chains.  `reduce(stage, overall)` returns the sums over
    ALL chains of all ranks ([d][6] for stage 1, [d][2] for stage 2 given the overall means [d][2]); `acov_sums(lag0, n_lags)`
    the pooled lag sums [d][n_lags]."""
    err = []

    def rcb(_user, stage, h_in, h_out):
        try:
            ov = np.ctypeslib.as_array(h_in, shape=(d, 2)).copy() if stage == 2 else None
            a = np.ascontiguousarray(reduce(int(stage), ov), dtype=np.float64)
            C.memmove(h_out, a.ctypes.data, d * (6 if stage == 1 else 2) * 8)
            return 0
        except Exception as ex:      # pragma: no cover
            err.append(ex)
            return FG_E_BAD_ARG

    def acb(_user, lag0, n_lags, out):
        try:
            a = np.ascontiguousarray(acov_sums(int(lag0), int(n_lags)), dtype=np.float64)
            C.memmove(out, a.ctypes.data, d * n_lags * 8)
            return 0
        except Exception as ex:      # pragma: no cover
            err.append(ex)
            return FG_E_BAD_ARG
    rfn, afn = REDUCE_FN(rcb), ACOV_FN(acb)
    rhat, ess, mean, std = (np.zeros(d) for _ in range(4))
    rc = lib().fg_diag_combine_reduced(int(m), int(n), int(d), rfn, afn, None, _dp(rhat), _dp(ess), _dp(mean), _dp(std))
    if err:
        raise err[0]
    _check(rc)
    return dict(r_hat=rhat, ess=ess, mean=mean, std=std)
