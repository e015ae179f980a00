"""Host-side sanitizer runs (CPU only; GPU AddressSanitizer is not available on the pool): the model-language parser, the program
builder / compiler, the run-time code generator (fg_jit.cpp: both generated translation units of every program that compiles) and
the diagnostics combination built with -fsanitize=address,undefined and driven through the C ABI with the
DSL sources of tests/dsl_models.py, malformed / truncated / mutated variants of them and random token streams; the oracle's
known-answer tests on its `make asan` build."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fugue_amd", "csrc")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=23", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=24")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    out = tmp_path_factory.mktemp("san") / "san_driver"
    sys.path.insert(0, ROOT)
    from fugue_amd import build as B
    B._write_jit_embed()                                       # fg_jit.cpp's embedded header text (a generated file)
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-DFG_BUILD", "-DFG_JIT_NO_HIP",
           "-Wno-unknown-pragmas", os.path.join(ROOT, "tests", "cpp", "san_driver.cpp")] + \
          [os.path.join(CSRC, f) for f in ("fg_program.cpp", "fg_dsl.cpp", "fg_diag_host.cpp", "fg_jit.cpp")] + ["-ldl", "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(out)


def _run(driver, *args):
    r = subprocess.run([driver, *args], capture_output=True, text=True, errors="replace", timeout=600, env=ENV)
    assert r.returncode == 0, f"sanitizer report (exit {r.returncode}):\n{r.stderr[-4000:]}"
    return r.stdout


def _records():
    """(source, data) pairs: every model of tests/dsl_models.py, the static-error sources of tests/test_dsl_cpu.py, and mutations."""
    from tests import dsl_models as Dm
    recs = []
    for name, val in vars(Dm).items():
        if isinstance(val, str) and name.isupper() and not name.endswith("_DATA") and ("pure(" in val or "sample(" in val):
            data = getattr(Dm, name + "_DATA", None)
            recs.append((val, data if isinstance(data, str) else (json.dumps(data) if data is not None else "<null>")))
    assert len(recs) >= 5
    base = list(recs)
    rng = np.random.default_rng(0)
    junk = ["", "pure(", "pure(nope)", 'let x <- sample(addr!("x"), Normal(0, 1)); pure(frob(x))', 'let s = "abc; pure(0)', "for i in 0..1000000000 { } pure(0)",
            'let x <- sample(addr!("x", 99999999999999999999), Normal(0,1)); pure(x)', "let x = 1e99999; pure(x)", "pure(((((((((((((((((((((((((((((((1)))))))))))))))))))))))))))))))",
            "(" * 5000 + "1" + ")" * 5000, 'let x <- sample(addr!("x"), Categorical(' + ",".join(["0.01"] * 100) + ")); pure(x)", "\x00\x01\x02 pure(0)", "pure(0) \xff\xfe"]
    for j in junk:
        recs.append((j, "<null>"))
    for src, data in base:
        for _ in range(25):                                       # truncations, deletions, duplications, byte flips
            b = bytearray(src.encode())
            op = rng.integers(4)
            if op == 0: b = b[:rng.integers(1, len(b))]
            elif op == 1: i = rng.integers(len(b) - 1); del b[i:i + rng.integers(1, 8)]
            elif op == 2: i = rng.integers(len(b) - 1); b[i:i] = b[i:i + rng.integers(1, 20)]
            else: b[rng.integers(len(b))] = rng.integers(32, 127)
            recs.append((b.decode(errors="replace").replace("\r", " "), data))
        for bad in ("[", "{", '{"y": [1, 2', '{"y": "text"}', "[1e999, -1e999, null]", '{"x": [], "y": []}', "[" + "1," * 5000 + "1]", "nul"):
            recs.append((src, bad))
    return recs


def test_dsl_front_end_under_asan_ubsan(driver, tmp_path):
    recs = _records()
    f = tmp_path / "records.txt"
    with open(f, "w", errors="replace") as fh:
        for src, data in recs:
            src = "\n".join(ln for ln in src.split("\n") if ln.strip() not in ("SRC", "DATA", "END"))
            fh.write("SRC\n" + src + "\nDATA\n" + data.replace("\n", " ") + "\nEND\n")
    out = _run(driver, "dsl", str(f))
    last = out.strip().splitlines()[-1]
    n_ok, n = int(last.split()[1]), int(last.split()[3])
    assert n == len(recs) and 5 <= n_ok < n                          # the intact models compile, the broken ones are refused -- none of them trips a sanitizer


def test_program_builder_under_asan_ubsan(driver):
    built = refused = 0
    for seed in range(1, 6):
        last = _run(driver, "program", str(seed)).strip().splitlines()[-1].split()
        built += int(last[2]); refused += int(last[4])
    assert built > 50 and refused > 50                                # both the accepting and the refusing paths ran


def test_diagnostics_combination_under_asan_ubsan(driver):
    for seed in (1, 2, 3):
        assert "rc=0" in _run(driver, "diag", str(seed))


def test_oracle_kats_on_its_asan_build(tmp_path):
    """`make asan` of the oracle + the reference's known-answer log-pdf values through it (tests/golden/reference_kats.json)."""
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = os.path.join(ROOT, "oracle", "libfugue_oracle_asan.so")
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(ENV, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0:exitcode=23", PYTHONMALLOC="malloc", FUGUE_ORACLE_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_kats.py"), os.path.join(ROOT, "tests", "test_oracle_behaviour.py"),
                        "-x", "-q", "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2500:], r.stderr[-2500:])
    assert " passed" in r.stdout
