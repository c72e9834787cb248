"""Runs minarrow_amd/cpp/ref_suite.bin: the reference's kernel tests (src/kernels/arithmetic/mod.rs:117-537,
src/kernels/bitmask/simd.rs:797-955) restated in C++ against the typed host mirror include/minarrow_hip.hpp,
which sits directly on the C ABI (Vec64 = pinned hipHostMalloc memory used in place by the kernels)."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BIN = ROOT / "minarrow_amd" / "cpp" / "ref_suite.bin"


def build():
    subprocess.run(["make", "-C", str(BIN.parent), "-s"], check=True)


@pytest.mark.gpu
def test_reference_suite_in_cpp():
    build()
    r = subprocess.run([str(BIN)], capture_output=True, text=True, timeout=300)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "0 failed" in r.stdout


def test_cpp_host_fails_loudly_without_a_gpu():
    from minarrow_amd import ffi

    if ffi.device_count() > 0:
        pytest.skip("a GPU is visible")
    build()
    r = subprocess.run([str(BIN)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "no HIP device is visible" in r.stdout
