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
