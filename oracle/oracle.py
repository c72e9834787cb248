"""ctypes loader for the C oracle (oracle/minarrow_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (minarrow_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "_build" / "libminarrow_oracle.so"
REF_INSPECT_PATH = _HERE / "_ref" / "libcinspect_arrow.so"

_lib = None


def build(force: bool = False) -> None:
    """Compiles the oracle (and oracle/_ref when /root/reference is present) with gcc."""
    src = _HERE / "minarrow_oracle.c"
    if force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-s"], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(LIB_PATH))
        _declare(_lib)
    return _lib


def _p(a):
    """Address of a numpy array (or None)."""
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _declare(l: C.CDLL) -> None:
    vp, sz, i32, i64, u64, f64 = C.c_void_p, C.c_size_t, C.c_int, C.c_int64, C.c_uint64, C.c_double
    sig = {
        "mo_sum_scalar_i64": (i64, [vp, sz]),
        "mo_sum_scalar_f64": (f64, [vp, sz]),
        "mo_simd_sum_i64": (i64, [vp, sz, i32]),
        "mo_simd_sum_f64": (f64, [vp, sz, i32]),
        "mo_simd_sum_f64_unrolled4": (f64, [vp, sz, i32]),
        "mo_simd_sum_i64_unrolled4": (i64, [vp, sz, i32]),
        "mo_chunked_sum_i64": (i64, [vp, sz, sz, i32]),
        "mo_chunked_sum_f64": (f64, [vp, sz, sz, i32]),
        "mo_par_sum_i64": (i64, [vp, sz, sz, i32, i32]),
        "mo_par_sum_f64": (f64, [vp, sz, sz, i32, i32]),
        "mo_par_fill_iota": (None, [vp, sz, i64, i32, i32]),
        "mo_masked_sum_i64": (None, [vp, sz, vp, sz, vp, vp]),
        "mo_masked_sum_i32": (None, [vp, sz, vp, sz, vp, vp]),
        "mo_masked_sum_f64": (None, [vp, sz, vp, sz, vp, vp]),
    }
    for name, (ret, args) in sig.items():
        if hasattr(l, name):
            fn = getattr(l, name)
            fn.restype = ret
            fn.argtypes = args


# ---- sums ------------------------------------------------------------------------------------------

def sum_scalar(a: np.ndarray):
    l = lib()
    if a.dtype == np.int64:
        return int(l.mo_sum_scalar_i64(_p(a), a.size))
    if a.dtype == np.float64:
        return float(l.mo_sum_scalar_f64(_p(a), a.size))
    raise TypeError(a.dtype)


def simd_sum(a: np.ndarray, lanes: int = 4):
    l = lib()
    if a.dtype == np.int64:
        return int(l.mo_simd_sum_i64(_p(a), a.size, lanes))
    if a.dtype == np.float64:
        return float(l.mo_simd_sum_f64(_p(a), a.size, lanes))
    raise TypeError(a.dtype)


def simd_sum_unrolled4(a: np.ndarray, lanes: int = 4):
    l = lib()
    if a.dtype == np.int64:
        return int(l.mo_simd_sum_i64_unrolled4(_p(a), a.size, lanes))
    if a.dtype == np.float64:
        return float(l.mo_simd_sum_f64_unrolled4(_p(a), a.size, lanes))
    raise TypeError(a.dtype)


def chunked_sum(a: np.ndarray, chunk: int = 1 << 20, lanes: int = 4):
    l = lib()
    if a.dtype == np.int64:
        return int(l.mo_chunked_sum_i64(_p(a), a.size, chunk, lanes))
    if a.dtype == np.float64:
        return float(l.mo_chunked_sum_f64(_p(a), a.size, chunk, lanes))
    raise TypeError(a.dtype)


def par_sum(a: np.ndarray, chunk: int = 1 << 20, lanes: int = 4, threads: int = 1):
    l = lib()
    if a.dtype == np.int64:
        return int(l.mo_par_sum_i64(_p(a), a.size, chunk, lanes, threads))
    if a.dtype == np.float64:
        return float(l.mo_par_sum_f64(_p(a), a.size, chunk, lanes, threads))
    raise TypeError(a.dtype)


def par_fill_iota(a: np.ndarray, start: int = 0, threads: int = 1) -> None:
    lib().mo_par_fill_iota(_p(a), a.size, start, 1 if a.dtype == np.float64 else 0, threads)


def masked_sum(a: np.ndarray, bits: np.ndarray, bit_offset: int = 0):
    """(sum, valid_count) — build-defined semantics (no reference implementation)."""
    l = lib()
    cnt = C.c_uint64()
    if a.dtype == np.int64:
        out = C.c_int64()
        l.mo_masked_sum_i64(_p(a), a.size, _p(bits), bit_offset, C.addressof(out), C.addressof(cnt))
    elif a.dtype == np.int32:
        out = C.c_int64()
        l.mo_masked_sum_i32(_p(a), a.size, _p(bits), bit_offset, C.addressof(out), C.addressof(cnt))
    elif a.dtype == np.float64:
        out = C.c_double()
        l.mo_masked_sum_f64(_p(a), a.size, _p(bits), bit_offset, C.addressof(out), C.addressof(cnt))
    else:
        raise TypeError(a.dtype)
    return out.value, int(cnt.value)
