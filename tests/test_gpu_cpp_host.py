"""Runs tests/cpp/ref_suite.bin: the reference's kernel tests (src/kernels/arithmetic/mod.rs:117-537,
src/kernels/bitmask/simd.rs:797-955) restated in C++ against the typed host mirror include/minarrow_hip.hpp,
which sits directly on the C ABI (Vec64 = pinned hipHostMalloc memory used in place by the kernels)."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BIN = ROOT / "tests" / "cpp" / "ref_suite.bin"


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


def _build_example(tmp_path, name="chunked_column"):
    exe = tmp_path / name
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", f"-I{ROOT / 'include'}", str(ROOT / "examples" / f"{name}.c"),
                    f"-L{ROOT / 'minarrow_amd' / 'lib'}", "-lminarrow_hip", f"-Wl,-rpath,{ROOT / 'minarrow_amd' / 'lib'}",
                    "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", str(exe)], check=True)
    return exe


@pytest.mark.gpu
def test_c_example_of_the_chunked_regime(tmp_path):
    """examples/chunked_column.c: a C99 host, the header and the .so — resident sum, the same column as 8192-row chunks
    (ma_sum_chunks), chunk + scalar for all chunks in one launch, consolidate; every result against its closed form."""
    exe = _build_example(tmp_path)
    r = subprocess.run([str(exe), str((1 << 24) + 12345)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout + r.stderr


def test_c_example_builds_and_fails_loudly_without_a_gpu(tmp_path):
    from minarrow_amd import ffi

    exe = _build_example(tmp_path)
    if ffi.device_count() > 0:
        pytest.skip("a GPU is visible")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "no HIP device is visible" in r.stdout


@pytest.mark.gpu
def test_c_example_of_the_partitioned_sums_with_bounded_waits(tmp_path):
    """examples/partitioned_sum.c: a C99 host over every visible GPU — the group with the overlapped RCCL exchange, the
    self-test first, a stepping loop whose waits are bounded; the third step's exchange is stalled with the library's fault
    hook: the host gets an error naming the pending member after its deadline, rebuilds the exchange one notch down on the same
    members and columns, and every step's totals (on every GPU) match the closed forms."""
    exe = _build_example(tmp_path, "partitioned_sum")
    r = subprocess.run([str(exe), str((1 << 22) + 77), "6", "300"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "self-test: PASS" in r.stdout and "did not finish within 300 ms" in r.stdout and "one notch down" in r.stdout
    assert r.stdout.strip().splitlines()[-1].startswith("ok: 6 steps"), r.stdout


@pytest.mark.gpu
@pytest.mark.rehearsal
def test_rehearsal_c_host_with_eight_members_on_one_gpu(tmp_path):
    """The same C99 host with EIGHT members on this box's one GPU through the loopback collective double: the self-test with
    peers, a stalled last member whose seven peers' all-gathers really wait for it, the error naming it after the deadline, the
    rebuild one notch down, every member's totals against the closed forms."""
    exe = _build_example(tmp_path, "partitioned_sum")
    r = subprocess.run([str(exe), str((1 << 22) + 77), "6", "400", "8"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "group of 8 GPU(s)" in r.stdout and "REHEARSAL" in r.stdout and "self-test: PASS: 8 members on 1 device(s)" in r.stdout, r.stdout
    assert "did not finish within 400 ms" in r.stdout and "member 7 " in r.stdout and "one notch down" in r.stdout, r.stdout
    assert r.stdout.strip().splitlines()[-1].startswith("ok: 6 steps"), r.stdout


@pytest.mark.gpu
def test_c_example_of_the_hot_loop_of_sums_on_one_stream_and_on_two_scan_lanes(tmp_path):
    """examples/hot_loop_sums.c: the reference's hot loop (one sum call per pass over the same arrays) from a C99 host — every
    pass one fused launch on the context's stream, then the same passes through ma_scan_lanes_*; every pass's record against the
    closed forms."""
    exe = _build_example(tmp_path, "hot_loop_sums")
    r = subprocess.run([str(exe), str((1 << 22) + 77), "40"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().splitlines()[-1].startswith("ok: 40 passes"), r.stdout


def test_c_hot_loop_example_builds_and_fails_loudly_without_a_gpu(tmp_path):
    from minarrow_amd import ffi

    exe = _build_example(tmp_path, "hot_loop_sums")
    if ffi.device_count() > 0:
        pytest.skip("a GPU is visible")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "no HIP device is visible" in r.stdout


def test_c_partitioned_example_builds_and_fails_loudly_without_a_gpu(tmp_path):
    from minarrow_amd import ffi

    exe = _build_example(tmp_path, "partitioned_sum")
    if ffi.device_count() > 0:
        pytest.skip("a GPU is visible")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "no HIP device is visible" in r.stdout
