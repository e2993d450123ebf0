"""The C++ host classes (include/nvstrings/NVStrings.h, NVCategory.h, NVText.h --
the reference's class names over the C ABI) compile against the library; without a
GPU they raise std::runtime_error, on the GPU box they pass the reference's gtest
known answers (tests/cpp/test_hostapi.cpp)."""
import os
import subprocess

import pytest

import cpulibs

CPP = os.path.join(cpulibs.ROOT, "tests", "cpp")
BIN = os.path.join(CPP, "test_hostapi")


def _build():
    subprocess.run(["make", "-s", "-C", CPP], check=True)


def test_cpp_hostapi_builds_and_fails_loudly_without_gpu():
    _build()
    out = subprocess.run([BIN, "nogpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_hostapi_reference_known_answers():
    _build()
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all host-API tests passed" in out.stdout
