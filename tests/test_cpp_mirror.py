"""The C++ host-side mirror (include/fugue_amd.hpp) compiled with g++ against the in-tree library."""
import os
import subprocess

import pytest

from fugue_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    E.lib()
    exe = str(tmp_path / "test_mirror")
    lib_dir = os.path.dirname(E.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_mirror.cpp"),
                    "-L", lib_dir, "-lfugue_amd", f"-Wl,-rpath,{lib_dir}", "-o", exe], check=True)
    return exe


def test_cpp_mirror_builds_programs_on_cpu(tmp_path):
    import torch
    exe = _build(tmp_path)
    args = [exe] if torch.cuda.is_available() else [exe, "--expect-no-device"]
    r = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "C++ mirror OK" in r.stdout


@pytest.mark.gpu
def test_cpp_mirror_runs_inference_on_gpu(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "--gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "C++ mirror OK" in r.stdout


def test_constant_division_matches_ieee_division(tmp_path):
    """fg_div_const (Markstein's sequence with RN(1/b)) == a / b bit for bit on 2e7 random and adversarial quotients."""
    exe = str(tmp_path / "test_div_const")
    subprocess.run(["g++", "-std=c++17", "-O2", "-mfma", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "test_div_const.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches 0" in r.stdout


def test_fast_log_and_sincos_are_within_one_ulp(tmp_path):
    """fg_fast_log / fg_fast_sincos (the normal generators' transcendentals) against libm on 4e6 generator inputs."""
    exe = str(tmp_path / "test_fast_math")
    subprocess.run(["g++", "-std=c++17", "-O2", "-mfma", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "test_fast_math.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fast math ok" in r.stdout


def test_log_density_derivatives_match_difference_quotients(tmp_path):
    """fg_dlogpdf / fg_digamma (the opt-in analytic gradients of the 17 distributions) against Richardson-extrapolated central
    differences of fg_logpdf on ~4e4 random (family, parameters, value, direction) draws inside the supports."""
    exe = str(tmp_path / "test_dlogpdf")
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "test_dlogpdf.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "directional derivatives" in r.stdout
