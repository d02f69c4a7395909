"""ctcdecode_amd/csrc/stl_emul.h (the nth_element / sort restatement a GPU lane runs) against the real libstdc++."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stl_emulation_matches_libstdcxx():
    exe = os.path.join(ROOT, "oracle", "_build", "stl_emul_check")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "native", "stl_emul_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe, "1500"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches=0" in r.stdout
