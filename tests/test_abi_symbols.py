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
    assert lib.ma_abi_version() == 1


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
    cpp_file.write_text('#include "minarrow_hip.hpp"\nint main() { return sizeof(ma::Bitmask) ? 0 : 1; }\n')
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", f"-I{inc}", str(cpp_file)], check=True)


def test_library_exports_nothing_but_the_c_abi():
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", str(ffi.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    declared = set(ffi.parse_header())
    assert exported == declared, (sorted(exported - declared)[:5], sorted(declared - exported)[:5])
