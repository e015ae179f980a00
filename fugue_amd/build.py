"""Builds the product library fugue_amd/lib/libfugue_amd.so for gfx950 (MI355X).

hipcc cross-compiles the device code without a GPU.  -ffp-contract=off: the reference (Rust) never fuses a*b+c, and the
parity tolerances in tests/ assume the same rounding.  Every translation unit is compiled to its own object (in
parallel; an object is rebuilt only when its source, a header or the flags changed) and the objects are linked into
the shared library; FG_LIB_PATH selects another output path (experiment builds never overwrite the product library).
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.environ.get("FG_LIB_PATH") or os.path.join(LIB_DIR, "libfugue_amd.so")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
SOURCES = ["fg_program.cpp", "fg_dsl.cpp", "fg_diag_host.cpp", "fg_engine.hip", "fg_hmc_sep.hip", "fg_hmc_lin.hip", "fg_hmc_interp.hip", "fg_mh.hip", "fg_mh_interp.hip", "fg_smc.hip", "fg_diag.hip", "fg_state.hip"]
HEADERS = ["fg_ir.h", "fg_math.h", "fg_interp.h", "fg_program.h", "fg_engine_internal.h", "fg_gradstream.h", "fg_cold.h",
           os.path.join("..", "..", "include", "fugue_amd.h")]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-DFG_BUILD", "-Wall",
         "-Wno-unused-function"]
LINK_LIBS = []


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def _sources():
    return [f for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]


def _extra_flags(extra=()):
    extra = list(extra)
    if os.environ.get("FG_EXTRA_DEFS"):
        extra += ["-D" + d for d in os.environ["FG_EXTRA_DEFS"].split(",")]
    if os.environ.get("FG_MIN_WAVES"):
        extra += ["-DFG_MIN_WAVES=" + os.environ["FG_MIN_WAVES"]]
    return extra


def _stamp(src: str, flags) -> str:
    h = hashlib.sha256()
    h.update(" ".join(flags).encode())
    for f in [os.path.join(CSRC, src)] + [os.path.join(CSRC, x) for x in HEADERS if os.path.exists(os.path.join(CSRC, x))] + [os.path.abspath(__file__)]:
        h.update(f.encode())
        h.update(str(os.path.getmtime(f)).encode())
    return h.hexdigest()


def _obj_paths(src: str, flags):
    tag = hashlib.sha256(" ".join(flags).encode()).hexdigest()[:10]
    base = os.path.join(OBJ_DIR, f"{os.path.splitext(src)[0]}.{tag}")
    return base + ".o", base + ".stamp"


def _obj_fresh(src: str, flags) -> bool:
    obj, st = _obj_paths(src, flags)
    try:
        return os.path.exists(obj) and open(st).read() == _stamp(src, flags)
    except OSError:
        return False


def needs_build(extra=()) -> bool:
    flags = FLAGS + _extra_flags(extra)
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in _sources()] + [os.path.join(CSRC, h) for h in HEADERS if os.path.exists(os.path.join(CSRC, h))] + [os.path.abspath(__file__)]
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    try:                                                   # a library built with other flags (an experiment) is stale for this build
        return open(LIB + ".flags").read() != " ".join(flags)
    except OSError:
        return True


def _compile(src: str, flags, verbose: bool):
    obj, st = _obj_paths(src, flags)
    tmp = f"{obj}.tmp.{os.getpid()}"
    cmd = [hipcc()] + flags + ["-c", "-x", "hip", os.path.join(CSRC, src), "-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        return src, r.stdout + r.stderr
    os.replace(tmp, obj)
    with open(st, "w") as f:
        f.write(_stamp(src, flags))
    return src, None


def build(force: bool = False, verbose: bool = False, extra=()) -> str:
    if not force and not needs_build(extra):
        return LIB
    flags = FLAGS + _extra_flags(extra)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = [s for s in _sources() if force or not _obj_fresh(s, flags)]
    jobs = max(1, min(len(todo), int(os.environ.get("FG_BUILD_JOBS", "6"))))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(lambda s: _compile(s, flags, verbose), todo))
    errs = [(s, e) for s, e in results if e]
    if errs:
        for s, e in errs:
            sys.stderr.write(f"---- {s} ----\n{e}\n")
        raise RuntimeError("hipcc failed building libfugue_amd.so")
    tmp = f"{LIB}.tmp.{os.getpid()}"                      # several ranks may build at once: write aside, then rename atomically
    objs = [_obj_paths(s, flags)[0] for s in _sources()]
    cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950"] + objs + LINK_LIBS + ["-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed linking libfugue_amd.so")
    os.replace(tmp, LIB)
    with open(LIB + ".flags", "w") as f:
        f.write(" ".join(flags))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
