"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU and exports exactly the
C ABI that include/minarrow_hip.h declares; with no device the product fails loudly instead of falling back."""
import ctypes as C
import re

from minarrow_amd import ffi


def test_header_prototypes_parse():
    protos = ffi.parse_header()
    assert len(protos) >= 40
    assert "ma_i64_sum" in protos and "ma_ctx_create" in protos
    ret, args = protos["ma_i64_sum"]
    assert ret == "ma_status" and [a[1] for a in args] == [
        "ctx", "data", "n", "mask_bits", "mask_bit_offset", "null_count", "out_sum", "out_valid_count"]


def test_library_exports_every_declared_symbol():
    lib = ffi.load_library()
    missing = [name for name in ffi.parse_header() if not hasattr(lib, name)]
    assert not missing, missing
    assert lib.ma_abi_version() == ffi.header_abi_version() == 5


def test_every_entry_point_cites_the_reference():
    """include/*.h must say which reference interface each group of entry points replaces (file:line)."""
    text = ffi.HEADER_PATH.read_text()
    cites = re.findall(r"(?:src|benches|tests)/[\w/\.]+\.(?:rs|c):\d+", text)
    assert len(cites) >= 10, cites


def test_no_gpu_means_loud_failure_not_fallback():
    lib = ffi.load_library()
    if lib.ma_device_count() > 0:
        return  # on a GPU box this property is vacuous
    h = C.c_void_p()
    status = lib.ma_ctx_create(0, C.byref(h))
    assert status == ffi.MA_ERR_NO_DEVICE and not h.value
    assert b"no CPU fallback" in lib.ma_last_error_string()
    p = C.c_void_p()
    assert lib.ma_alloc64_pinned(64, C.byref(p)) == ffi.MA_ERR_NO_DEVICE
    assert ffi.status_name(status) == lib.ma_status_name(status).decode()


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under minarrow_amd/ may reference it."""
    root = ffi.LIB_PATH.parent.parent
    offenders = []
    for path in list(root.rglob("*.py")) + list(root.rglob("*.hip")) + list(root.rglob("*.hpp")) + list(root.rglob("*.cpp")):
        text = path.read_text(errors="ignore")
        if re.search(r"\boracle\b", text) and path.name != "__init__.py":
            offenders.append(str(path))
        if re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M):
            offenders.append(str(path))
    assert not offenders, offenders


def test_header_is_plain_c_and_the_cpp_mirror_compiles(tmp_path):
    """The boundary header must be consumable by a C compiler (cgo / bindgen / JNI all start there)."""
    import subprocess

    inc = ffi.HEADER_PATH.parent
    c_file = tmp_path / "use.c"
    c_file.write_text('#include "minarrow_hip.h"\nint main(void) { return ma_abi_version() == MA_ABI_VERSION ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", f"-I{inc}", str(c_file)],
                   check=True)
    cpp_file = tmp_path / "use.cpp"
    cpp_file.write_text('#include "minarrow_hip.hpp"\n#include "minarrow_hip_routing.hpp"\n#include "minarrow_hip_parallel.hpp"\n'
                        'int main() { return sizeof(ma::Bitmask) && sizeof(ma::Group) && ma::row_chunks(130, 2)[1].first == 64 ? 0 : 1; }\n')
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", f"-I{inc}", str(cpp_file)], check=True)


def test_library_exports_nothing_but_the_c_abi():
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", str(ffi.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    declared = set(ffi.parse_header()) | set(ffi.parse_header(ffi.TESTING_HEADER_PATH))
    assert exported == declared, (sorted(exported - declared)[:5], sorted(declared - exported)[:5])


def test_fault_hooks_live_apart_from_the_product_abi_and_are_inert_by_default():
    """The fault hooks (stall / corrupt / fail an exchange, pretend a member lives elsewhere, hold a scan, the Power series) are
    declared in include/minarrow_hip_testing.h, not in the product header and not in the Rust binding, and do nothing unless
    MINARROW_HIP_TEST_HOOKS=1 was set when the library was loaded."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    product, testing = set(ffi.parse_header()), set(ffi.parse_header(ffi.TESTING_HEADER_PATH))
    assert not product & testing
    assert not [n for n in product if "_test_" in n or n.startswith("ma_test_")], "a hook is declared in the product header"
    assert all("_test_" in n or n.startswith("ma_test_") for n in testing) and len(testing) == 9
    assert "TESTING ONLY" not in ffi.HEADER_PATH.read_text()
    binding = (Path(__file__).resolve().parent.parent / "bindings" / "minarrow_hip_sys.rs").read_text()
    assert not [n for n in testing if n in binding]
    # a NULL handle is refused as such only when the hooks are live; by default "disabled" comes first... but argument checks are
    # allowed to come first: use the one hook that needs no handle state — every hook reports MA_ERR_UNSUPPORTED when disabled
    code = ("import ctypes as C; from minarrow_amd import ffi; l = ffi.load_library(); "
            "st = l.ma_test_pow_series(1, 0, None, None, 0); print(l.ma_test_hooks_enabled(), st, l.ma_last_error_string().decode())")
    root = str(Path(__file__).resolve().parent.parent)
    off = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=root,
                         env={k: v for k, v in os.environ.items() if k != "MINARROW_HIP_TEST_HOOKS"})
    assert off.returncode == 0, off.stderr
    enabled, status, message = off.stdout.strip().split(" ", 2)
    assert (enabled, int(status)) == ("0", ffi.MA_ERR_UNSUPPORTED) and "MINARROW_HIP_TEST_HOOKS=1" in message
    on = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=root,
                        env=dict(os.environ, MINARROW_HIP_TEST_HOOKS="1"))
    assert on.returncode == 0 and on.stdout.split()[:2] == ["1", "0"], on.stdout + on.stderr  # n == 0: nothing to do, MA_OK


def test_rust_binding_file_covers_every_entry_point():
    """bindings/minarrow_hip_sys.rs (generated by tools/gen_rust_ffi.py) declares exactly the header's functions."""
    import re
    from pathlib import Path

    from minarrow_amd import ffi

    text = (Path(__file__).resolve().parent.parent / "bindings" / "minarrow_hip_sys.rs").read_text()
    declared = set(re.findall(r"pub fn (ma_\w+)\(", text))
    assert declared == set(ffi.parse_header().keys())
    assert "*const *const c_void" in text and "-> MaStatus" in text


def test_bench_self_launch_modes_fail_loudly_without_gpus():
    """`python bench.py --gpus N` started directly must take the one-process group path (no launcher needed) and, on a
    box without N GPUs, say so and exit 2 before touching torch; a launcher / --gpus mismatch is refused the same way."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    lib = ffi.load_library()
    if lib.ma_device_count() >= 2:
        return  # only meaningful where the GPUs are missing
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr and r.stdout.strip() == ""
    env["WORLD_SIZE"] = "4"
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr


def test_environment_surface_is_read(tmp_path):
    """MINARROW_HIP_MIN_ROWS / MINARROW_HIP_DEVICES are read by the library (no GPU needed for these two)."""
    import os
    import subprocess
    import sys

    code = ("from minarrow_amd import ffi; l = ffi.load_library(); "
            "print(l.ma_min_device_rows(), l.ma_device_count(), l.ma_min_device_rows_for(1), l.ma_min_device_rows_for(2))")
    env = dict(os.environ, MINARROW_HIP_MIN_ROWS="12345", MINARROW_HIP_DEVICES="0", MINARROW_HIP_MIN_ROWS_ELEMENTWISE="777")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120,
                       cwd=str(ffi.LIB_PATH.parent.parent.parent))
    assert r.returncode == 0, r.stderr
    rows, devices, elementwise, scan_bits = r.stdout.split()
    assert rows == "12345" and int(devices) <= 1 and elementwise == "777" and scan_bits == str(1 << 21)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                       env={k: v for k, v in os.environ.items() if not k.startswith("MINARROW_HIP_")},
                       cwd=str(ffi.LIB_PATH.parent.parent.parent))
    # defaults = the measured crossovers of a ~12 us synchronous call against one host thread
    assert r.stdout.split()[0] == str(1 << 18) and r.stdout.split()[2] == str(1 << 15)


def test_environment_table_is_generated_from_the_source():
    """INTEGRATION.md section 5's table comes from minarrow_amd/csrc/ma_env.hpp (tools/gen_env_table.py): every MINARROW_HIP_* variable
    the product's sources read is documented there, nothing documented is unread, and the committed table is the generated one."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "gen_env_table.py"), "--check"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr


def test_the_collective_library_can_be_named_and_a_stand_in_says_what_it_is():
    """MINARROW_HIP_RCCL_PATH: the library opens the collective library the host names and no other — a path that does not open is an
    error (no silent fall-back to the system's RCCL), and the loopback collective double of tests/loopback_rccl (test infrastructure:
    the twelve entry points load_rccl() resolves, liveness semantics on one device) is reported as a REHEARSAL by ma_rccl_path and
    ma_rccl_version. No GPU needed: nothing is initialised by opening it."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    double = root / "tests" / "loopback_rccl" / "libloopback_rccl.so"
    assert double.exists(), f"{double} is missing: make -C tests/cpp"
    out = subprocess.run(["nm", "-D", "--defined-only", str(double)], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    resolved = {"ncclGetVersion", "ncclGetUniqueId", "ncclCommInitRank", "ncclCommInitAll", "ncclCommDestroy", "ncclCommCount", "ncclCommAbort",
                "ncclAllGather", "ncclAllReduce", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"}  # ma_rccl.hip: load_rccl()
    assert resolved | {"ncclLoopbackDoubleInfo"} <= exported, sorted(resolved - exported)
    code = "from minarrow_amd import ffi; l = ffi.load_library(); print(repr(l.ma_rccl_path().decode()), l.ma_rccl_version())"
    env = {k: v for k, v in os.environ.items() if k != "MINARROW_HIP_RCCL_PATH"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=str(root),
                       env=dict(env, MINARROW_HIP_RCCL_PATH=str(double)))
    assert r.returncode == 0, r.stderr
    path, version = r.stdout.strip().rsplit(" ", 1)
    assert path.startswith("'REHEARSAL") and str(double) in path and version == "9900", r.stdout
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=str(root),
                       env=dict(env, MINARROW_HIP_RCCL_PATH="/nonexistent/librccl.so"))
    assert r.returncode == 0 and r.stdout.strip() == "'' 0", r.stdout + r.stderr  # an error, not the system's RCCL
