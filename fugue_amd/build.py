"""Builds the product library fugue_amd/lib/libfugue_amd.so for gfx950 (MI355X).

hipcc cross-compiles the device code without a GPU.  -ffp-contract=off: the reference (Rust)
never fuses a*b+c, and the parity tolerances in tests/ assume the same rounding.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libfugue_amd.so")
SOURCES = ["fg_program.cpp", "fg_dsl.cpp", "fg_engine.hip", "fg_smc.hip", "fg_diag.hip"]
HEADERS = ["fg_ir.h", "fg_math.h", "fg_interp.h", "fg_program.h", "fg_engine_internal.h", "fg_gradstream.h", os.path.join("..", "..", "include", "fugue_amd.h")]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
         "-fgpu-rdc" if False else "-DFG_BUILD", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra=()) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs = []
    for f in SOURCES:
        p = os.path.join(CSRC, f)
        srcs += (["-x", "hip", p] if f.endswith((".hip", ".cpp")) else [p])
    if os.environ.get("FG_EXTRA_DEFS"):
        extra = list(extra) + ["-D" + d for d in os.environ["FG_EXTRA_DEFS"].split(",")]
    if os.environ.get("FG_MIN_WAVES"):
        extra = list(extra) + ["-DFG_MIN_WAVES=" + os.environ["FG_MIN_WAVES"]]
    tmp = f"{LIB}.tmp.{os.getpid()}"                      # several ranks may build at once: write aside, then rename atomically
    cmd = [hipcc()] + FLAGS + list(extra) + srcs + ["-o", tmp]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed building libfugue_amd.so")
    os.replace(tmp, LIB)
    if verbose and r.stderr:
        sys.stderr.write(r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
