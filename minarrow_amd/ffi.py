"""ctypes binding of libminarrow_hip.so — the C ABI declared in include/minarrow_hip.h.

This is the same binding a Rust `extern "C"` block (or cgo / JNI stub) would make; see INTEGRATION.md.
There is no CPU implementation behind it: if the shared library has not been built, or no GPU is visible,
the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path

_ROOT = Path(__file__).resolve().parent
# MINARROW_HIP_LIB: another build of the same library — the tuning build (make -C minarrow_amd/csrc TUNING=1 -> build/tuning/), whose
# ma_ctx_set_variant takes every bit; tools/ sweeps and the variant A/B tests use it. Unset: the shipped in-tree library.
LIB_PATH = Path(os.environ["MINARROW_HIP_LIB"]) if os.environ.get("MINARROW_HIP_LIB") else _ROOT / "lib" / "libminarrow_hip.so"
HEADER_PATH = _ROOT.parent / "include" / "minarrow_hip.h"
TESTING_HEADER_PATH = _ROOT.parent / "include" / "minarrow_hip_testing.h"  # the fault hooks: inert unless MINARROW_HIP_TEST_HOOKS=1 at load

MA_OK = 0
MA_ERR_LENGTH_MISMATCH = 1
MA_ERR_DIVIDE_BY_ZERO = 2
MA_ERR_UNSUPPORTED = 3
MA_ERR_INVALID_ARGUMENT = 4
MA_ERR_DEVICE = 5
MA_ERR_NO_DEVICE = 6

# C type names used in the header -> ctypes
_CTYPES = {
    "void": None,
    "int32_t": C.c_int32,
    "uint32_t": C.c_uint32,
    "int64_t": C.c_int64,
    "uint64_t": C.c_uint64,
    "size_t": C.c_size_t,
    "float": C.c_float,
    "double": C.c_double,
    "ma_status": C.c_int32,
    "int8_t": C.c_int8,
    "uint8_t": C.c_uint8,
    "int16_t": C.c_int16,
    "uint16_t": C.c_uint16,
}


class MinarrowHipError(RuntimeError):
    """Raised for any non-OK ma_status. `.status` holds the code, mirroring KernelError's variants
    (src/enums/error.rs:157-187 of the reference)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"{status_name(status)}: {message}")
        self.status = status
        self.message = message


class LibraryNotBuilt(RuntimeError):
    pass


def parse_header(path: Path = HEADER_PATH):
    """Returns {function name: (return type string, [(type string, arg name), ...])} for every prototype in
    the header. Used to set ctypes signatures and by the tests that check the exported symbol set."""
    text = path.read_text()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#[^\n]*$", " ", text, flags=re.M)  # preprocessor lines
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(ma_\w+)\s*\(([^;{}()]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef") or ret.startswith("#"):
            continue
        arglist = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                arglist.append((mm.group(1).strip(), mm.group(2)))
        protos[name] = (ret, arglist)
    return protos


def header_abi_version(path: Path = HEADER_PATH) -> int:
    """MA_ABI_VERSION as the header states it: what a binding generated from this header expects ma_abi_version() to be."""
    return int(re.search(r"^#define\s+MA_ABI_VERSION\s+(\d+)", path.read_text(), flags=re.M).group(1))


def _to_ctype(tname: str):
    t = tname.replace("const", " ").replace("struct", " ")
    t = " ".join(t.split())
    stars = t.count("*")
    base = t.replace("*", "").strip()
    if stars == 0:
        return _CTYPES[base]
    # every pointer crosses the boundary as a raw address
    return C.c_void_p


_lib = None


def load_library() -> C.CDLL:
    """Loads (once) the in-tree shared library and applies the header's signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise LibraryNotBuilt(
            f"{LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C minarrow_amd/csrc`. There is no CPU fallback."
        )
    lib = C.CDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2))
    lib.ma_abi_version.restype = C.c_int32
    if lib.ma_abi_version() != header_abi_version():
        raise LibraryNotBuilt(f"{LIB_PATH} has ABI version {lib.ma_abi_version()}, include/minarrow_hip.h declares "
                              f"{header_abi_version()}: rebuild it (make -C minarrow_amd/csrc)")
    for name, (ret, args) in {**parse_header(), **parse_header(TESTING_HEADER_PATH)}.items():
        fn = getattr(lib, name)
        if ret.replace(" ", "") == "constchar*":
            fn.restype = C.c_char_p
        elif "*" in ret:
            fn.restype = C.c_void_p
        else:
            fn.restype = _to_ctype(ret)
        fn.argtypes = [_to_ctype(t) for t, _ in args]
    _lib = lib
    return lib


def status_name(status: int) -> str:
    names = {
        0: "MA_OK",
        1: "MA_ERR_LENGTH_MISMATCH",
        2: "MA_ERR_DIVIDE_BY_ZERO",
        3: "MA_ERR_UNSUPPORTED",
        4: "MA_ERR_INVALID_ARGUMENT",
        5: "MA_ERR_DEVICE",
        6: "MA_ERR_NO_DEVICE",
    }
    return names.get(status, f"MA_ERR_{status}")


def check(status: int) -> None:
    if status != MA_OK:
        msg = load_library().ma_last_error_string()
        raise MinarrowHipError(status, msg.decode() if msg else "")


def device_count() -> int:
    return int(load_library().ma_device_count())
