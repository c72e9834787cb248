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


def _host_cpu_flags() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                return " ".join(sorted(line.split(":", 1)[1].split()))
    except OSError:
        pass
    return "unknown"


def build(force: bool = False) -> None:
    """Compiles the oracle (and oracle/_ref when /root/reference is present) with gcc. The library is built with
    -march=native, and a prebuilt copy travels with the repo to the GPU box: it is rebuilt there when that host's CPU
    feature set differs from the build host's (an AVX-512 build must not run on a host without it)."""
    src = _HERE / "minarrow_oracle.c"
    stamp = _HERE / "_build" / "build_host_cpu_flags.txt"
    flags = _host_cpu_flags()
    same_host = stamp.exists() and stamp.read_text() == flags
    if force or not same_host or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        if LIB_PATH.exists():
            LIB_PATH.unlink()  # make must not consider it up to date
        subprocess.run(["make", "-C", str(_HERE), "-s"], check=True)
        stamp.parent.mkdir(parents=True, exist_ok=True)
        stamp.write_text(flags)


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
        **{f"mo_masked_sum_{t}": (None, [vp, sz, vp, sz, vp, vp]) for t in ("i8", "u8", "i16", "u16", "u32")},
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


def set_pool_pinning(on: bool) -> None:
    """Pool workers created from now on stay on one CPU each of the set the process may run on (bench.py's cpu_baseline)."""
    lib().mo_pool_set_pinning(1 if on else 0)


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
    elif a.dtype in (np.int8, np.uint8, np.int16, np.uint16, np.uint32):
        # extended_numeric_types: 64-bit wrapping sum of the widened elements; bits=None = every row valid
        out = C.c_int64()
        name = {np.dtype(np.int8): "i8", np.dtype(np.uint8): "u8", np.dtype(np.int16): "i16", np.dtype(np.uint16): "u16",
                np.dtype(np.uint32): "u32"}[a.dtype]
        getattr(l, f"mo_masked_sum_{name}")(_p(a), a.size, _p(bits) if bits is not None else None, bit_offset,
                                            C.addressof(out), C.addressof(cnt))
        if a.dtype.kind == "u":
            return out.value & ((1 << 64) - 1), int(cnt.value)
    else:
        raise TypeError(a.dtype)
    return out.value, int(cnt.value)


# ---- elementwise arithmetic / FMA / bitmask kernels ---------------------------------------------------

OPS = {"add": 0, "subtract": 1, "multiply": 2, "divide": 3, "remainder": 4, "power": 5, "floordiv": 6}
OK, LENGTH_MISMATCH, PANIC_DIV_ZERO, PANIC_OVERFLOW = 0, 1, 2, 4

_TAGS = {
    np.dtype(np.int8): "i8", np.dtype(np.int16): "i16", np.dtype(np.int32): "i32", np.dtype(np.int64): "i64",
    np.dtype(np.uint8): "u8", np.dtype(np.uint16): "u16", np.dtype(np.uint32): "u32", np.dtype(np.uint64): "u64",
    np.dtype(np.float32): "f32", np.dtype(np.float64): "f64",
}
# build.rs:67-110 lane tables: AVX-512 / AVX2 / SSE2+NEON
LANES = {"avx512": {1: 64, 2: 32, 4: 16, 8: 8}, "avx2": {1: 32, 2: 16, 4: 8, 8: 4}, "sse2": {1: 16, 2: 8, 4: 4, 8: 2}}


def tag(dtype) -> str:
    return _TAGS[np.dtype(dtype)]


def aligned_empty(n: int, dtype, align: int = 64, offset_bytes: int = 0) -> np.ndarray:
    """Uninitialised array whose data pointer is `offset_bytes` past an `align`-byte boundary
    (Vec64 gives 64-byte alignment, src/lib.rs:99; offset_bytes != 0 forces the scalar fallback)."""
    dt = np.dtype(dtype)
    raw = np.empty(n * dt.itemsize + align + offset_bytes + 64, dtype=np.uint8)
    start = (-raw.ctypes.data) % align + offset_bytes
    return raw[start:start + n * dt.itemsize].view(dt)


def aligned_copy(a: np.ndarray, align: int = 64, offset_bytes: int = 0) -> np.ndarray:
    out = aligned_empty(a.size, a.dtype, align, offset_bytes)
    out[:] = a
    return out


def pack_bits(valid) -> np.ndarray:
    """bool sequence -> Arrow validity bytes (LSB first), padded so whole-u64-word reads stay in bounds."""
    valid = np.asarray(valid, dtype=bool)
    packed = np.packbits(valid, bitorder="little")
    return pad_bits(packed, valid.size)


def pad_bits(bits: np.ndarray, len_bits: int, extra_words: int = 2) -> np.ndarray:
    n_bytes = ((len_bits + 63) // 64 + extra_words) * 8
    out = aligned_empty(max(n_bytes, 8), np.uint8)
    out[:] = 0
    m = min(bits.size, out.size)
    out[:m] = bits[:m]
    return out


def unpack_bits(bits: np.ndarray, n: int, offset: int = 0) -> np.ndarray:
    return np.unpackbits(np.ascontiguousarray(bits), bitorder="little")[offset:offset + n].astype(bool)


def _declare_kernels(l: C.CDLL) -> None:
    vp, sz, i32, u64 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64
    for t in ("i8", "i16", "i32", "i64", "u8", "u16", "u32", "u64"):
        getattr(l, f"mo_int_dense_std_{t}").argtypes = [i32, vp, vp, vp, sz]
        getattr(l, f"mo_int_masked_std_{t}").argtypes = [i32, vp, vp, vp, vp, vp, sz]
        getattr(l, f"mo_int_dense_simd_{t}").argtypes = [i32, vp, vp, vp, sz, i32]
        getattr(l, f"mo_int_masked_simd_{t}").argtypes = [i32, vp, vp, vp, sz, vp, vp, sz, i32]
        getattr(l, f"mo_apply_int_{t}").argtypes = [vp, sz, vp, sz, i32, vp, sz, vp, vp, i32, vp]
    for t in ("f32", "f64"):
        getattr(l, f"mo_float_dense_{t}").argtypes = [i32, vp, vp, vp, sz]
        getattr(l, f"mo_float_masked_std_{t}").argtypes = [i32, vp, vp, vp, vp, vp, sz]
        getattr(l, f"mo_float_masked_simd_{t}").argtypes = [i32, vp, vp, vp, sz, vp, vp, sz, i32]
        getattr(l, f"mo_apply_float_{t}").argtypes = [vp, sz, vp, sz, i32, vp, sz, vp, vp, i32, vp]
        getattr(l, f"mo_fma_dense_{t}").argtypes = [vp, vp, vp, vp, sz]
        getattr(l, f"mo_fma_dense_{t}").restype = None
        getattr(l, f"mo_fma_masked_{t}").argtypes = [vp, vp, vp, vp, vp, vp, sz]
        getattr(l, f"mo_fma_masked_{t}").restype = None
        getattr(l, f"mo_apply_fma_{t}").argtypes = [vp, sz, vp, sz, vp, sz, vp, vp, vp, i32]
    l.mo_bitmask_new_set_all.argtypes = [vp, sz, i32]
    l.mo_bitmask_new_set_all.restype = None
    l.mo_bitmask_count_ones.argtypes = [vp, sz]
    l.mo_bitmask_count_ones.restype = sz
    l.mo_simd_mask_bits.argtypes = [vp, sz, sz, sz, i32]
    l.mo_simd_mask_bits.restype = u64
    l.mo_write_mask_bits.argtypes = [vp, sz, u64, i32]
    l.mo_write_mask_bits.restype = None
    for name in ("mo_all_true_mask_simd", "mo_all_false_mask_simd"):
        getattr(l, name).argtypes = [vp, sz, i32]
    for name in ("mo_all_true_mask_std", "mo_all_false_mask_std"):
        getattr(l, name).argtypes = [vp, sz]
    l.mo_bitmask_binop.argtypes = [i32, vp, sz, vp, sz, sz, vp]
    l.mo_bitmask_binop.restype = None
    l.mo_bitmask_not.argtypes = [vp, sz, sz, vp]
    l.mo_bitmask_not.restype = None
    l.mo_bitmask_in.argtypes = [vp, sz, vp, sz, sz, vp]
    l.mo_bitmask_in.restype = None
    l.mo_bitmask_not_in.argtypes = [vp, sz, vp, sz, sz, vp, vp]
    l.mo_bitmask_not_in.restype = None
    for name in ("mo_bitmask_eq", "mo_bitmask_ne"):
        getattr(l, name).argtypes = [vp, sz, vp, sz, sz, vp]
    for name in ("mo_bitmask_all_eq", "mo_bitmask_all_ne"):
        getattr(l, name).argtypes = [vp, sz, vp, sz, sz]
    l.mo_bitmask_popcount.argtypes = [vp, sz, sz]
    l.mo_bitmask_popcount.restype = sz
    l.mo_merge_bitmasks.argtypes = [vp, vp, sz, vp]
    l.mo_bitmask_union.argtypes = [vp, vp, sz, vp]
    l.mo_bitmask_union.restype = None
    for t, ct in (("u8", C.c_uint8), ("u16", C.c_uint16), ("u32", C.c_uint32), ("u64", C.c_uint64)):
        fn = getattr(l, f"mo_simd_eq_mask_{t}")
        fn.argtypes = [vp, sz, ct, ct, vp]
        fn.restype = None


_kernels_declared = False


def klib() -> C.CDLL:
    global _kernels_declared
    l = lib()
    if not _kernels_declared:
        _declare_kernels(l)
        _kernels_declared = True
    return l


def _opcode(op) -> int:
    return OPS[op] if isinstance(op, str) else int(op)


def _mask_words(n: int) -> np.ndarray:
    out = aligned_empty(((n + 63) // 64 + 2) * 8, np.uint8)
    out[:] = 0
    return out


def apply_int(lhs: np.ndarray, rhs: np.ndarray, op, mask: np.ndarray | None = None, mask_len: int | None = None,
              lanes: int = 8):
    """apply_int_<t> (src/kernels/arithmetic/dispatch.rs:65-133). Inputs are used where they lie: 64-byte
    aligned inputs take the SIMD body, anything else the scalar body, as in the reference.
    Returns (status, out, out_mask_bits or None, used_simd)."""
    t = tag(lhs.dtype)
    out = aligned_empty(lhs.size, lhs.dtype)
    out[:] = 0
    out_mask = _mask_words(lhs.size) if mask is not None else None
    used = C.c_int(0)
    st = getattr(klib(), f"mo_apply_int_{t}")(_p(lhs), lhs.size, _p(rhs), rhs.size, _opcode(op), _p(mask),
                                              lhs.size if mask_len is None else mask_len, _p(out), _p(out_mask), lanes,
                                              C.addressof(used))
    return st, out, out_mask, bool(used.value)


def apply_float(lhs: np.ndarray, rhs: np.ndarray, op, mask: np.ndarray | None = None, mask_len: int | None = None,
                lanes: int = 8):
    """apply_float_<t> (dispatch.rs:138-206). Returns (status, out, out_mask_bits or None, used_simd)."""
    t = tag(lhs.dtype)
    out = aligned_empty(lhs.size, lhs.dtype)
    out[:] = 0
    out_mask = _mask_words(lhs.size) if mask is not None else None
    used = C.c_int(0)
    st = getattr(klib(), f"mo_apply_float_{t}")(_p(lhs), lhs.size, _p(rhs), rhs.size, _opcode(op), _p(mask),
                                                lhs.size if mask_len is None else mask_len, _p(out), _p(out_mask),
                                                lanes, C.addressof(used))
    return st, out, out_mask, bool(used.value)


def apply_fma(lhs, rhs, acc, mask=None, force_unfused: bool = False):
    """apply_fma_<t> (dispatch.rs:211-290). Returns (status, out, out_mask_bits or None)."""
    t = tag(lhs.dtype)
    out = aligned_empty(lhs.size, lhs.dtype)
    out[:] = 0
    out_mask = _mask_words(lhs.size) if mask is not None else None
    st = getattr(klib(), f"mo_apply_fma_{t}")(_p(lhs), lhs.size, _p(rhs), rhs.size, _p(acc), acc.size, _p(mask), _p(out),
                                              _p(out_mask), 1 if force_unfused else 0)
    return st, out, out_mask


def int_body(kind: str, lhs, rhs, op, mask=None, mask_len=None, lanes: int = 8):
    """One of the four integer bodies by name: dense_std | masked_std | dense_simd | masked_simd."""
    t = tag(lhs.dtype)
    n = lhs.size
    out = aligned_empty(n, lhs.dtype)
    out[:] = 0
    l = klib()
    if kind == "dense_std":
        return getattr(l, f"mo_int_dense_std_{t}")(_opcode(op), _p(lhs), _p(rhs), _p(out), n), out, None
    if kind == "dense_simd":
        return getattr(l, f"mo_int_dense_simd_{t}")(_opcode(op), _p(lhs), _p(rhs), _p(out), n, lanes), out, None
    out_mask = _mask_words(n)
    l.mo_bitmask_new_set_all(_p(out_mask), n, 1)
    if kind == "masked_std":
        st = getattr(l, f"mo_int_masked_std_{t}")(_opcode(op), _p(lhs), _p(rhs), _p(mask), _p(out), _p(out_mask), n)
    else:
        st = getattr(l, f"mo_int_masked_simd_{t}")(_opcode(op), _p(lhs), _p(rhs), _p(mask), n if mask_len is None else mask_len,
                                                   _p(out), _p(out_mask), n, lanes)
    return st, out, out_mask


def float_body(kind: str, lhs, rhs, op, mask=None, mask_len=None, lanes: int = 8):
    t = tag(lhs.dtype)
    n = lhs.size
    out = aligned_empty(n, lhs.dtype)
    out[:] = 0
    l = klib()
    if kind == "dense":
        return getattr(l, f"mo_float_dense_{t}")(_opcode(op), _p(lhs), _p(rhs), _p(out), n), out, None
    out_mask = _mask_words(n)
    l.mo_bitmask_new_set_all(_p(out_mask), n, 1)
    if kind == "masked_std":
        st = getattr(l, f"mo_float_masked_std_{t}")(_opcode(op), _p(lhs), _p(rhs), _p(mask), _p(out), _p(out_mask), n)
    else:
        st = getattr(l, f"mo_float_masked_simd_{t}")(_opcode(op), _p(lhs), _p(rhs), _p(mask), n if mask_len is None else mask_len,
                                                     _p(out), _p(out_mask), n, lanes)
    return st, out, out_mask


LOGICAL = {"and": 0, "or": 1, "xor": 2}


def bitmask_binop(op, lhs, lhs_off, rhs, rhs_off, n):
    out = _mask_words(n)
    klib().mo_bitmask_binop(LOGICAL[op] if isinstance(op, str) else op, _p(lhs), lhs_off, _p(rhs), rhs_off, n, _p(out))
    return out


def bitmask_not(src, off, n):
    out = _mask_words(n)
    klib().mo_bitmask_not(_p(src), off, n, _p(out))
    return out


def bitmask_in(lhs, lhs_off, rhs, rhs_off, n):
    out = _mask_words(n)
    klib().mo_bitmask_in(_p(lhs), lhs_off, _p(rhs), rhs_off, n, _p(out))
    return out


def bitmask_not_in(lhs, lhs_off, rhs, rhs_off, n):
    out, scratch = _mask_words(n), _mask_words(n)
    klib().mo_bitmask_not_in(_p(lhs), lhs_off, _p(rhs), rhs_off, n, _p(out), _p(scratch))
    return out


def bitmask_eq(a, ao, b, bo, n, negate: bool = False):
    """(panics, out_bits)"""
    out = _mask_words(n)
    fn = klib().mo_bitmask_ne if negate else klib().mo_bitmask_eq
    return bool(fn(_p(a), ao, _p(b), bo, n, _p(out))), out


def bitmask_all_eq(a, ao, b, bo, n) -> int:
    return int(klib().mo_bitmask_all_eq(_p(a), ao, _p(b), bo, n))


def bitmask_all_ne(a, ao, b, bo, n) -> int:
    return int(klib().mo_bitmask_all_ne(_p(a), ao, _p(b), bo, n))


def bitmask_popcount(bits, off, n) -> int:
    return int(klib().mo_bitmask_popcount(_p(bits), off, n))


def all_true(bits, n, lanes: int | None = 8) -> bool:
    l = klib()
    return bool(l.mo_all_true_mask_std(_p(bits), n) if lanes is None else l.mo_all_true_mask_simd(_p(bits), n, lanes))


def all_false(bits, n, lanes: int | None = 8) -> bool:
    l = klib()
    return bool(l.mo_all_false_mask_std(_p(bits), n) if lanes is None else l.mo_all_false_mask_simd(_p(bits), n, lanes))


def merge_bitmasks(l_bits, r_bits, n):
    out = _mask_words(n)
    some = klib().mo_merge_bitmasks(_p(l_bits), _p(r_bits), n, _p(out))
    return out if some else None


def bitmask_union(l_bits, r_bits, n):
    out = _mask_words(n)
    klib().mo_bitmask_union(_p(l_bits), _p(r_bits), n, _p(out))
    return out


def simd_eq_mask(data: np.ndarray, field_mask: int, target: int):
    out = _mask_words(data.size)
    getattr(klib(), f"mo_simd_eq_mask_{tag(data.dtype)}")(_p(data), data.size, field_mask, target, _p(out))
    return out


def count_ones(bits, n) -> int:
    return int(klib().mo_bitmask_count_ones(_p(bits), n))


def simd_mask_bits(bits, mask_len, offset, n, lanes) -> int:
    return int(klib().mo_simd_mask_bits(_p(bits), mask_len, offset, n, lanes))


def write_mask_bits(out_bits, offset, mbits, lanes) -> None:
    klib().mo_write_mask_bits(_p(out_bits), offset, mbits, lanes)


def consolidate_column(chunks, masks=None, mask_offsets=None):
    """(values, validity bits or None) — src/traits/consolidate.rs:80-207."""
    l = klib()
    k = len(chunks)
    dt = chunks[0].dtype
    total = sum(c.size for c in chunks)
    out = np.zeros(total, dtype=dt)
    out_mask = _mask_words(total)
    data_arr = (C.c_void_p * k)(*[c.ctypes.data for c in chunks])
    len_arr = (C.c_size_t * k)(*[c.size for c in chunks])
    mask_arr = (C.c_void_p * k)(*[(m.ctypes.data if m is not None else None) for m in masks]) if masks is not None else None
    off_arr = (C.c_size_t * k)(*mask_offsets) if mask_offsets is not None else None
    l.mo_consolidate_column.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]
    has = l.mo_consolidate_column(dt.itemsize, k, C.cast(data_arr, C.c_void_p), C.cast(len_arr, C.c_void_p),
                                  C.cast(mask_arr, C.c_void_p) if mask_arr is not None else None,
                                  C.cast(off_arr, C.c_void_p) if off_arr is not None else None, _p(out), _p(out_mask))
    return out, (out_mask if has else None)


def consolidate_boolean_column(chunks, masks=None):
    """chunks: [(bits uint8 array, bit offset, len)], masks: same shape or None entries.
    Returns (data bits, validity bits or None), each 8*ceil(total/64) bytes — bitmask.rs:520-592,
    boolean.rs:627-653."""
    l = klib()
    k = len(chunks)
    total = sum(c[2] for c in chunks)
    out = np.zeros(((total + 63) // 64) * 8 + 8, dtype=np.uint8)
    out_mask = np.zeros_like(out)
    arr = lambda vals, ty: (ty * k)(*vals)
    bits_arr = arr([c[0].ctypes.data for c in chunks], C.c_void_p)
    bytes_arr = arr([c[0].size for c in chunks], C.c_size_t)
    off_arr = arr([c[1] for c in chunks], C.c_size_t)
    len_arr = arr([c[2] for c in chunks], C.c_size_t)
    if masks is None:
        masks = [None] * k
    m_arr = arr([(m[0].ctypes.data if m is not None else None) for m in masks], C.c_void_p)
    mb_arr = arr([(m[0].size if m is not None else 0) for m in masks], C.c_size_t)
    mo_arr = arr([(m[1] if m is not None else 0) for m in masks], C.c_size_t)
    l.mo_consolidate_boolean_column.argtypes = [C.c_size_t] + [C.c_void_p] * 9
    l.mo_consolidate_boolean_column.restype = C.c_int
    cast = lambda a: C.cast(a, C.c_void_p)
    has = l.mo_consolidate_boolean_column(k, cast(bits_arr), cast(bytes_arr), cast(off_arr), cast(len_arr), cast(m_arr),
                                          cast(mb_arr), cast(mo_arr), _p(out), _p(out_mask))
    nbytes = ((total + 63) // 64) * 8
    return out[:nbytes], (out_mask[:nbytes] if has else None)


# ---- arena consolidation (src/structs/arena.rs) --------------------------------------------------------------------

def align64(n: int) -> int:
    """src/utils.rs:178-180."""
    return (n + 63) & ~63


def arena_regions(regions):
    """Cursor rule of Arena::push_slice / reserve_slice (arena.rs:152-232): every region starts at the next 64-byte
    boundary and the cursor ends right after it. regions: [(elem_size, count)] -> (byte offsets, used bytes)."""
    cursor, offsets = 0, []
    for elem, count in regions:
        cursor = align64(cursor)
        offsets.append(cursor)
        cursor += elem * count
    return offsets, cursor


def arena_capacity_for_regions(entries) -> int:
    """Arena::capacity_for_regions (arena.rs:442-447): entries are (len, elem_size)."""
    return sum(align64(n * e) for n, e in entries)


def consolidate_table_arena(columns):
    """consolidate_tables_arena for numeric columns (arena.rs:1187-1296 capacity pass, :1298-1340 write pass,
    Arena::write_slices :264-308). columns: [(chunks, masks or None, mask_offsets or None)], every column with the
    same batch row counts. Returns (arena bytes [capacity], data offsets, mask offsets (None = column without nulls),
    used bytes). Per column in order: data region of n_rows elements, then — iff any batch of the column has a mask —
    a validity region of ceil(n_rows / 8) bytes; batches without a mask contribute all-valid bits."""
    n_rows = sum(c.size for c in columns[0][0])
    mask_bytes = (n_rows + 7) // 8
    capacity, regions, has_nulls = 0, [], []
    for chunks, masks, _ in columns:
        nulls = masks is not None and any(m is not None for m in masks)
        has_nulls.append(nulls)
        elem = chunks[0].dtype.itemsize
        capacity += align64(n_rows * elem)
        regions.append((elem, n_rows))
        if nulls:
            capacity += align64(mask_bytes)
            regions.append((1, mask_bytes))
    offsets, used = arena_regions(regions)
    arena = np.zeros(capacity, dtype=np.uint8)  # Arena::with_capacity pre-fills with zeros (arena.rs:125-130)
    data_offsets, mask_offsets, it = [], [], iter(offsets)
    for (chunks, masks, mask_offs), nulls in zip(columns, has_nulls):
        values, bits = consolidate_column(chunks, masks if nulls else None, mask_offs if nulls else None)
        off = next(it)
        data_offsets.append(off)
        raw = values.view(np.uint8)
        arena[off:off + raw.size] = raw
        if nulls:
            moff = next(it)
            mask_offsets.append(moff)
            arena[moff:moff + mask_bytes] = bits[:mask_bytes]
        else:
            mask_offsets.append(None)
    return arena, data_offsets, mask_offsets, used
