"""Builds and runs tests/cpp/test_host_mirror.cpp: the reference's tests restated in C++ against the C++ host mirror
(include/kofft_hip.hpp) of kofft's FftImpl / RealFftImpl / stft interface, linked to libkofft_hip.so."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
EXE = ROOT / "tests" / "cpp" / "test_host_mirror"


def build():
    from oracle import pyoracle

    pyoracle.build()
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", str(ROOT / "tests/cpp/test_host_mirror.cpp"), "-o", str(EXE),
           f"-L{ROOT / 'kofft_amd/lib'}", "-lkofft_hip", f"-L{ROOT / 'oracle'}", "-lkofft_oracle",
           f"-Wl,-rpath,{ROOT / 'kofft_amd/lib'}", f"-Wl,-rpath,{ROOT / 'oracle'}", "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def test_cpp_host_mirror_compiles():
    """CPU: the header and the test translate and link against the C ABI (no GPU needed to build)."""
    build()
    assert EXE.exists()


@pytest.mark.gpu
def test_cpp_host_mirror_runs(oracle):
    build()
    res = subprocess.run([str(EXE)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert " 0 failed" in res.stdout
