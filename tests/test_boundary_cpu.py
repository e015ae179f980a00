"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol that
include/fugue_amd.h declares, the host-side program compiler orders sites like the reference's
BTreeMap, rejects what the reference panics on, and refuses to run without a GPU (no fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from fugue_amd import engine as E
from fugue_amd import model as M
from fugue_amd import workloads as W
from tests.models import ZOO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "fugue_amd.h")).read()
    declared = set(re.findall(r"\b(fg_[a-z0-9_]+)\s*\(", header))
    declared -= {"fg_tok"}
    lib = ctypes.CDLL(E.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(E.ABI_SYMBOLS), declared ^ set(E.ABI_SYMBOLS)
    assert E.lib().fg_abi_version() == 1


def test_rust_binding_declares_every_entry_point():
    """rust/fugue-gpu/src/ffi.rs (the reference-side binding, unverified source) stays in step with include/fugue_amd.h."""
    ffi = open(os.path.join(ROOT, "rust", "fugue-gpu", "src", "ffi.rs")).read()
    bound = set(re.findall(r"pub fn (fg_[a-z0-9_]+)\s*\(", ffi))
    assert bound == set(E.ABI_SYMBOLS), bound ^ set(E.ABI_SYMBOLS)


def test_rust_binding_and_ctypes_state_the_headers_abi():
    """Argument counts, pointer-ness / const-ness, integer widths and `#[repr(C)]` field order of rust/fugue-gpu/src/ffi.rs, and the
    ctypes declarations the tests call through, against include/fugue_amd.h -- parsed, not compiled (tests/abi_check.py)."""
    from tests import abi_check as A
    header = open(os.path.join(ROOT, "include", "fugue_amd.h")).read()
    ffi = open(os.path.join(ROOT, "rust", "fugue-gpu", "src", "ffi.rs")).read()
    parsed = A.parse_header(header)
    assert set(parsed["fns"]) == set(E.ABI_SYMBOLS) and len(parsed["structs"]) >= 7 and len(parsed["callbacks"]) == 2
    assert A.compare_header_rust(header, ffi) == []
    assert A.compare_header_ctypes(header, E.lib()) == []
    # the check has teeth: a swapped field, a dropped argument, a narrowed integer, a lost `const` and a swapped pair of arguments fail
    mutations = [
        ("pub n_leapfrog: i32, pub target_accept: f64,", "pub target_accept: f64, pub n_leapfrog: i32,", "struct fg_hmc_config"),
        ("pub fn fg_program_data(p: *mut fg_program, name: *const c_char, v: *const f64, n: i64) -> c_int;",
         "pub fn fg_program_data(p: *mut fg_program, name: *const c_char, v: *const f64) -> c_int;", "fg_program_data: 3 arguments"),
        ("v: *const f64, n: i64) -> c_int;", "v: *const f64, n: i32) -> c_int;", "fg_program_data: argument 3"),
        ("pub fn fg_program_n_sites(p: *const fg_program)", "pub fn fg_program_n_sites(p: *mut fg_program)", "fg_program_n_sites: argument 0"),
        ("pub log_evidence: f64, pub n_steps: i32, pub n_model_runs: i64", "pub log_evidence: f64, pub n_steps: i64, pub n_model_runs: i64", "struct fg_smc_result"),
        ("lo: i64, hi: i64) -> c_int;", "lo: i64, hi: u64) -> c_int;", "fg_program_sample_discrete_uniform: argument 3"),
    ]
    for old, new, expect in mutations:
        assert old in ffi, old
        bad = A.compare_header_rust(header, ffi.replace(old, new, 1))
        assert bad and any(expect in b for b in bad), (expect, bad)
    bad = A.compare_header_rust(header.replace("int fg_program_factor(fg_program *p, const fg_tok *toks, int n);",
                                               "int fg_program_factor(fg_program *p, int n, const fg_tok *toks);"), ffi)
    assert any("fg_program_factor" in b for b in bad)


@pytest.mark.parametrize("name", list(ZOO))
def test_site_order_is_lexicographic_and_matches_oracle(oracle, name):
    prog = ZOO[name]()
    cp = E.compile_model(prog)
    om = oracle.OracleModel(prog)
    assert cp.site_names == sorted(prog.sample_addresses(), key=lambda s: s.encode())
    assert cp.site_names == om.site_names and cp.site_vtypes == om.site_vtypes
    assert (cp.S, cp.d, cp.O) == (om.S, om.d, om.O)


def test_indexed_addresses_sort_as_strings():
    cp = E.compile_model(W.normal_sites(12))
    assert cp.site_names[:4] == ["x#0", "x#1", "x#10", "x#11"]      # "x#10" < "x#2" (address.rs:150-157)
    assert M.addr("a#b", "c\\d") == "a\\#b#c\\\\d"                   # escape_addr_segment (address.rs:189-223)


def test_duplicate_address_is_rejected():
    P = M.Program()
    P.sample(M.addr("x"), M.Normal(0, 1))
    P.sample(M.addr("x"), M.Normal(0, 1))
    with pytest.raises(M.FugueError) as ei:
        E.compile_model(P)
    assert ei.value.code == M.ErrorCode.AddressConflict           # 301, interpreters.rs:23-33


def test_constructor_validation_codes():
    for ctor, code in [(lambda: M.Normal(0, -1), 101), (lambda: M.Normal(float("nan"), 1), 100),
                       (lambda: M.Bernoulli(1.5), 102), (lambda: M.Uniform(2, 1), 103), (lambda: M.Beta(0, 1), 104),
                       (lambda: M.Gamma(1, 0), 105), (lambda: M.Categorical([0.5, 0.6]), 102),
                       (lambda: M.DiscreteUniform(3, 1), 103), (lambda: M.Poisson(0), 105)]:
        with pytest.raises(M.FugueError) as ei:
            ctor()
        assert ei.value.code == code


def test_structure_varying_model_is_refused():
    def model():
        return M.sample(M.addr("b"), M.Bernoulli(0.5)).bind(
            lambda b: M.sample(M.addr("x"), M.Normal(0, 1)) if b else M.pure(0.0))
    with pytest.raises(M.StructureError):
        M.trace_model(model)


def test_sparse_dependency_counts():
    cp = E.compile_model(W.normal_sites(32))
    assert cp.n_instructions == 64 and cp.dep_counts == [2] * 32      # own prior + own observation
    cp = E.compile_model(W.reference_model(8))                        # mu feeds every x#i prior
    mu = cp.site_names.index("mu")
    assert cp.dep_counts[cp.f64_sites.index(mu)] == 8 and min(cp.dep_counts) == 2


def test_engine_refuses_to_run_without_a_gpu():
    """The product has no CPU fallback: without a device engine creation fails loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(E.EngineError) as ei:
        E.Engine(E.compile_model(W.readme_normal()), 64, seed=1)
    assert "no CPU fallback" in str(ei.value)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fugue_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "fugue_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_stream_records_report_what_the_compiler_could_build():
    """fg_program_stream_records: (gradient records, score records, record kinds) -- 0 means the interpreter kernels."""
    from tests.models import ZOO
    want = {"readme": (2, 2, 0), "normal32": (64, 64, 0), "refmodel8": (22, 15, 0), "ridge": (100, 28, 1), "hier_scale": (29, 20, 2),
            "mixture": (44, 24, 2), "alldists": (0, 0, 0), "coin": (0, 0, 0)}
    for name, rec in want.items():
        assert E.compile_model(ZOO[name]()).stream_records == rec, name


@pytest.mark.parametrize("name", ["alldists", "poisson_glm", "hier_logsigma", "logistic", "mixture", "coin"])
def test_jit_source_compiles_for_gfx950(name):
    """fg_jit.cpp: the generated translation unit of a program's HMC kernel (one C++ statement per interpreter instruction behind
    fg_hmc_jit_body.h) goes through hiprtc for gfx950 -- no GPU needed to compile."""
    lib = E.lib()
    lib.fg_debug_jit_compile.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_longlong, ctypes.c_char_p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]
    cp = E.compile_model(ZOO[name]())
    src = ctypes.create_string_buffer(8 << 20); log = ctypes.create_string_buffer(1 << 20); n = ctypes.c_longlong()
    rc = lib.fg_debug_jit_compile(cp.h, src, len(src), log, len(log), ctypes.byref(n))
    assert rc == 0 and n.value > 0, log.value.decode()[:2000]
    text = src.value.decode()
    assert "fg_jit_task" in text and "fg_jit_score" in text and "k_hmc_jit_steps" in text


@pytest.mark.parametrize("name,W", [("refmodel8", 8), ("hier_scale", 4), ("alldists", 16)])
def test_jit_task_code_compiles_for_gfx950(name, W, monkeypatch):
    """The compiled HMC unit generated behind a task split (round 4: every wave's (coordinate, sign) tasks as straight-line code,
    fg_jit_wave_tasks; the program's d and S as literals) goes through hiprtc; every task appears exactly once."""
    lib = E.lib()
    lib.fg_debug_jit_compile.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_longlong, ctypes.c_char_p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]
    monkeypatch.setenv("FG_DEBUG_JIT_TASKS", str(W))
    monkeypatch.setenv("FG_DEBUG_JIT_COORDS", str(W))
    cp = E.compile_model(ZOO[name]())
    src = ctypes.create_string_buffer(8 << 20); log = ctypes.create_string_buffer(1 << 20); n = ctypes.c_longlong()
    rc = lib.fg_debug_jit_compile(cp.h, src, len(src), log, len(log), ctypes.byref(n))
    assert rc == 0 and n.value > 0, log.value.decode()[:2000]
    text = src.value.decode()
    assert f"#define FG_JIT_BAKED_W {min(W, 2 * cp.d)}" in text and f"#define FG_JIT_K_D {cp.d}\n" in text and f"#define FG_JIT_K_S {cp.S}\n" in text
    body = text[text.index("void fg_jit_wave_tasks("):text.index("#define FG_JIT_FUSED_W")]
    for t in range(2 * cp.d):
        assert body.count(f"ev[{t} * FG_WAVE] = ") == 1, t
    grad = text[text.index("bool fg_jit_wave_grad("):text.index("double fg_jit_task(")]      # whole coordinates per wave (the one-barrier gradient)
    for k in range(cp.d):
        assert grad.count(f"FG_JIT_GRAD_COORD({k}, ") == 1, k


@pytest.mark.parametrize("name", ["refmodel8", "mixture"])
def test_jit_mh_unit_with_the_launch_shape_as_literals_compiles(name, monkeypatch):
    """The multi-wave MH unit of a score-stream program with the launch shape as literals (FG_MHMW_K_*) and the control wave's in-order sums
    pinned side by side (the form an engine with one tile per CU gets) goes through hiprtc."""
    lib = E.lib()
    lib.fg_debug_jit_compile.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_longlong, ctypes.c_char_p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]
    for k, v in (("FG_DEBUG_JIT_MHMW", "1"), ("FG_DEBUG_JIT_NSEG", "8"), ("FG_DEBUG_JIT_BAKE", "4160"), ("FG_MH_SUMS_FORM", "4")): monkeypatch.setenv(k, v)
    cp = E.compile_model(ZOO[name]())
    src = ctypes.create_string_buffer(8 << 20); log = ctypes.create_string_buffer(1 << 20); n = ctypes.c_longlong()
    rc = lib.fg_debug_jit_compile(cp.h, src, len(src), log, len(log), ctypes.byref(n))
    assert rc == 0 and n.value > 0, log.value.decode()[:2000]
    text = src.value.decode()
    assert "#define FG_MHMW_K_W 8\n" in text and "#define FG_MHMW_K_EXP 4160\n" in text and 'asm volatile("" : "+v"(a), "+v"(b));' in text
