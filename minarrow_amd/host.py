"""Host-side convenience layer over the C ABI (minarrow_amd.ffi) used by the tests and bench.py.

It only moves addresses and sizes across the boundary; all arithmetic happens in the HIP kernels.
Buffers may be numpy arrays (pageable host memory: the library stages them), `DeviceBuffer`s,
`PinnedBuffer`s (the Vec64 stand-in), torch tensors, or raw integer addresses.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Optional, Tuple

import numpy as np

from . import ffi

_DTYPE_TAG = {
    np.dtype(np.int64): "i64",
    np.dtype(np.uint64): "u64",
    np.dtype(np.int32): "i32",
    np.dtype(np.uint32): "u32",
    np.dtype(np.float64): "f64",
    np.dtype(np.float32): "f32",
    np.dtype(np.int8): "i8",
    np.dtype(np.uint8): "u8",
    np.dtype(np.int16): "i16",
    np.dtype(np.uint16): "u16",
}
_TAG_DTYPE = {v: k for k, v in _DTYPE_TAG.items()}


def tag_of(dtype) -> str:
    return _DTYPE_TAG[np.dtype(dtype)]


def dtype_of(tag: str) -> np.dtype:
    return _TAG_DTYPE[tag]


def addr_of(x) -> int:
    """Raw address of anything buffer-like. None -> 0 (NULL)."""
    if x is None:
        return 0
    if isinstance(x, int):
        return x
    if isinstance(x, (DeviceBuffer, PinnedBuffer, Registered)):
        return x.ptr
    if isinstance(x, np.ndarray):
        if not x.flags["C_CONTIGUOUS"]:
            raise ValueError("numpy buffers passed across the C ABI must be C-contiguous")
        return x.ctypes.data
    if hasattr(x, "data_ptr"):  # torch tensor
        return int(x.data_ptr())
    raise TypeError(f"cannot take the address of {type(x)!r}")


class FusedColumn(C.Structure):
    """struct ma_fused_column (include/minarrow_hip.h)."""
    _fields_ = [("data", C.c_void_p), ("n", C.c_size_t), ("mask_bits", C.c_void_p), ("mask_bit_offset", C.c_size_t),
                ("null_count", C.c_int64), ("format_code", C.c_int32), ("reserved", C.c_int32), ("out", C.c_void_p)]


class DeviceBuffer:
    """Device-resident bytes owned through ma_dev_alloc / ma_dev_free. output=True: ma_dev_alloc_output — a block meant
    to be written by the kernels, picked for its write rate (`write_gbps` holds the measured figure, 0 if none)."""

    def __init__(self, ctx: "Context", nbytes: int, output: bool = False):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        self.write_gbps = 0.0
        p = C.c_void_p()
        if output:
            rate = C.c_float()
            ffi.check(ctx.lib.ma_dev_alloc_output(ctx.handle, self.nbytes, C.byref(p), C.byref(rate)))
            self.write_gbps = float(rate.value)
            ms, held, measured, considered, good = C.c_double(), C.c_size_t(), C.c_int32(), C.c_int32(), C.c_float()
            ffi.check(ctx.lib.ma_dev_alloc_output_stats(C.addressof(ms), C.addressof(held), C.addressof(measured),
                                                        C.addressof(considered), C.addressof(good)))
            # what the placement search cost: wall time, bytes held at its peak, blocks probed / looked at, the threshold
            self.alloc_stats = {"search_ms": float(ms.value), "candidates_held_bytes": int(held.value),
                                "blocks_measured": int(measured.value), "blocks_considered": int(considered.value),
                                "good_gbps": float(good.value)}
        else:
            ffi.check(ctx.lib.ma_dev_alloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = int(p.value)
        live = getattr(ctx, "_buffers", None)
        if live is not None:
            live.add(self)  # Context.close() returns whatever is still outstanding

    def offset(self, nbytes: int) -> int:
        return self.ptr + int(nbytes)

    def upload(self, arr: np.ndarray, byte_offset: int = 0) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        assert byte_offset + arr.nbytes <= self.nbytes
        ffi.check(self.ctx.lib.ma_dev_upload(self.ctx.handle, self.ptr + byte_offset, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, dtype, count: int, byte_offset: int = 0) -> np.ndarray:
        out = np.empty(int(count), dtype=dtype)
        assert byte_offset + out.nbytes <= self.nbytes
        ffi.check(self.ctx.lib.ma_dev_download(self.ctx.handle, out.ctypes.data, self.ptr + byte_offset, out.nbytes))
        return out

    def free(self) -> None:
        if self.ptr:
            self.ctx.lib.ma_dev_free(self.ctx.handle, self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            if self.ptr and self.ctx.handle:
                self.free()
        except Exception:
            pass


class PinnedBuffer:
    """64-byte aligned pinned host memory from ma_alloc64_pinned (hipHostMalloc): the Vec64 stand-in.
    `view(dtype)` exposes it to numpy without a copy; kernels read and write it in place."""

    def __init__(self, nbytes: int):
        self.lib = ffi.load_library()
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        ffi.check(self.lib.ma_alloc64_pinned(self.nbytes, C.byref(p)))
        self.ptr = int(p.value)

    def view(self, dtype, count: Optional[int] = None) -> np.ndarray:
        dt = np.dtype(dtype)
        count = self.nbytes // dt.itemsize if count is None else int(count)
        buf = (C.c_uint8 * (count * dt.itemsize)).from_address(self.ptr)
        buf._owner = self  # the array's base chain now holds the PinnedBuffer: no recycling under a live view
        return np.frombuffer(buf, dtype=dt, count=count)

    def free(self) -> None:
        if self.ptr:
            self.lib.ma_free_pinned(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Registered:
    """An existing host buffer (numpy array) pinned in place for its lifetime: ma_host_register / ma_host_unregister."""

    def __init__(self, arr: np.ndarray):
        self.arr = arr
        self.ptr = arr.ctypes.data
        ffi.check(ffi.load_library().ma_host_register(self.ptr, arr.nbytes))
        self._live = True

    def release(self) -> None:
        if self._live:
            self._live = False
            ffi.check(ffi.load_library().ma_host_unregister(self.ptr))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.release()


class Graph:
    """A recorded sequence of calls (hipGraph). `launch()` replays it on the owning context's stream."""

    def __init__(self, ctx: "Context", handle: int):
        self.ctx, self.handle = ctx, handle

    @property
    def nodes(self) -> int:
        n = C.c_size_t()
        ffi.check(self.ctx.lib.ma_graph_node_count(self.handle, C.addressof(n)))
        return int(n.value)

    def launch(self) -> None:
        ffi.check(self.ctx.lib.ma_graph_launch(self.ctx.handle, self.handle))

    def destroy(self) -> None:
        if self.handle:
            self.ctx.lib.ma_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Context:
    """One device + one HIP stream (ma_ctx)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = ffi.load_library()
        h = C.c_void_p()
        if stream is None:
            ffi.check(self.lib.ma_ctx_create(int(device), C.byref(h)))
        else:
            ffi.check(self.lib.ma_ctx_create_on_stream(int(device), int(stream), C.byref(h)))
        self.handle = h.value
        self.device = int(device)
        self._buffers = weakref.WeakSet()

    # -- lifecycle -------------------------------------------------------------------------------
    def close(self) -> None:
        if self.handle:
            for lanes in list(getattr(self, "_scan_lanes", ())):  # a pipeline goes before the context it borrows
                lanes.close()
            for buf in list(getattr(self, "_buffers", ())):  # device blocks that outlived their users
                buf.free()
            self.lib.ma_ctx_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def synchronize(self) -> None:
        ffi.check(self.lib.ma_ctx_synchronize(self.handle))

    def set_async(self, on: bool) -> None:
        ffi.check(self.lib.ma_ctx_set_async(self.handle, 1 if on else 0))

    def set_blocks_per_cu(self, n: int) -> None:
        ffi.check(self.lib.ma_ctx_set_blocks_per_cu(self.handle, int(n)))

    def set_staging_tile(self, tile_bytes: int) -> None:
        """Bytes of one operand per tile of the pipelined host-operand path (0 = stage whole operands)."""
        ffi.check(self.lib.ma_ctx_set_staging_tile(self.handle, int(tile_bytes)))

    def set_grid(self, workgroups: int) -> None:
        ffi.check(self.lib.ma_ctx_set_grid(self.handle, int(workgroups)))

    def set_variant(self, v: int) -> None:
        """ma_ctx_set_variant: FORM bits (FORM_BITS: they force one of two product paths, for tests) in every build; any other bit
        is a tuning form that only a library built with TUNING=1 has — the shipped one refuses it (MA_ERR_UNSUPPORTED)."""
        ffi.check(self.lib.ma_ctx_set_variant(self.handle, int(v)))

    @property
    def compute_units(self) -> int:
        return int(self.lib.ma_ctx_compute_units(self.handle))

    def timer_start(self) -> None:
        ffi.check(self.lib.ma_ctx_timer_start(self.handle))

    def timer_stop(self) -> None:
        ffi.check(self.lib.ma_ctx_timer_stop(self.handle))

    def timer_elapsed_ms(self) -> float:
        ms = C.c_float()
        ffi.check(self.lib.ma_ctx_timer_elapsed_ms(self.handle, C.byref(ms)))
        return float(ms.value)

    def mark(self, index: int) -> None:
        """Record timing mark `index` on the context's stream (ma_ctx_mark)."""
        ffi.check(self.lib.ma_ctx_mark(self.handle, index))

    def mark_elapsed_ms(self, from_index: int, to_index: int) -> float:
        ms = C.c_float()
        ffi.check(self.lib.ma_ctx_mark_elapsed_ms(self.handle, from_index, to_index, C.byref(ms)))
        return float(ms.value)

    # -- hipGraph capture ------------------------------------------------------------------------------
    def capture_begin(self) -> None:
        """Record (instead of run) every following call on this context until `capture_end`."""
        ffi.check(self.lib.ma_ctx_capture_begin(self.handle))

    def capture_end(self) -> "Graph":
        g = C.c_void_p()
        ffi.check(self.lib.ma_ctx_capture_end(self.handle, C.addressof(g)))
        return Graph(self, g.value)

    # -- memory ----------------------------------------------------------------------------------
    @property
    def hip_device(self) -> int:
        """The HIP runtime's ordinal of this context's device (`device` is the library ordinal: MINARROW_HIP_DEVICES)."""
        return int(self.lib.ma_ctx_hip_device(self.handle))

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def alloc_output(self, nbytes: int) -> DeviceBuffer:
        """A block the kernels will write (ma_dev_alloc_output): the fastest-writing of a few candidates."""
        return DeviceBuffer(self, nbytes, output=True)

    def to_device(self, arr: np.ndarray, pad_bytes: int = 0) -> DeviceBuffer:
        arr = np.ascontiguousarray(arr)
        buf = DeviceBuffer(self, arr.nbytes + pad_bytes)
        if pad_bytes:
            ffi.check(self.lib.ma_dev_memset(self.handle, buf.ptr, 0, buf.nbytes))
        if arr.nbytes:
            buf.upload(arr)
        return buf

    def dev_copy(self, dst, src, nbytes: int) -> None:
        """Device-to-device copy through the runtime (hipMemcpyAsync on the context's stream)."""
        ffi.check(self.lib.ma_dev_copy(self.handle, addr_of(dst), addr_of(src), int(nbytes)))

    def dev_memset(self, dst, byte_value: int, nbytes: int) -> None:
        ffi.check(self.lib.ma_dev_memset(self.handle, addr_of(dst), int(byte_value), int(nbytes)))

    # -- synthetic inputs ------------------------------------------------------------------------
    def synth_iota(self, tag: str, dst, n: int, start: int = 0) -> None:
        fn = getattr(self.lib, f"ma_synth_iota_{tag}")
        ffi.check(fn(self.handle, addr_of(dst), int(n), int(start)))

    def synth_splitmix(self, tag: str, dst, n: int, seed: int, first_index: int = 0) -> None:
        fn = getattr(self.lib, f"ma_synth_splitmix_{tag}")
        ffi.check(fn(self.handle, addr_of(dst), int(n), int(seed), int(first_index)))

    def synth_validity(self, dst, n_bits: int, seed: int, first_index: int = 0, null_every: int = 10) -> None:
        ffi.check(self.lib.ma_synth_validity(self.handle, addr_of(dst), int(n_bits), int(seed), int(first_index),
                                             int(null_every)))

    # -- reductions ------------------------------------------------------------------------------
    def sum(self, tag: str, data, n: int, mask=None, mask_bit_offset: int = 0, null_count: int = -1) -> Tuple:
        """(sum, valid_count). Integer sums come back as Python ints (wrapped to 64 bits), float sums as float."""
        fn = getattr(self.lib, f"ma_{tag}_sum")
        cnt = C.c_uint64()
        if tag in ("f64", "f32"):
            out = C.c_double()
        elif tag in ("i64", "i32", "i16", "i8"):
            out = C.c_int64()
        else:
            out = C.c_uint64()
        ffi.check(fn(self.handle, addr_of(data), int(n), addr_of(mask), int(mask_bit_offset), int(null_count),
                     C.addressof(out), C.addressof(cnt)))
        return out.value, int(cnt.value)

    def sum_dd(self, tag: str, data, n: int, mask=None, mask_bit_offset: int = 0, null_count: int = -1):
        """(hi, lo, valid_count) — unevaluated double-double sum for f64/f32."""
        fn = getattr(self.lib, f"ma_{tag}_sum_dd")
        hi, lo, cnt = C.c_double(), C.c_double(), C.c_uint64()
        ffi.check(fn(self.handle, addr_of(data), int(n), addr_of(mask), int(mask_bit_offset), int(null_count),
                     C.addressof(hi), C.addressof(lo), C.addressof(cnt)))
        return hi.value, lo.value, int(cnt.value)

    def mean(self, tag: str, data, n: int, mask=None, mask_bit_offset: int = 0, null_count: int = -1):
        fn = getattr(self.lib, f"ma_{tag}_mean")
        out, cnt = C.c_double(), C.c_uint64()
        ffi.check(fn(self.handle, addr_of(data), int(n), addr_of(mask), int(mask_bit_offset), int(null_count),
                     C.addressof(out), C.addressof(cnt)))
        return out.value, int(cnt.value)

    def sum_into(self, tag: str, data, n: int, out_sum, out_count=None, mask=None, mask_bit_offset: int = 0,
                 null_count: int = -1, dd_lo=None) -> None:
        """Enqueue-style variant: results are written to caller-provided (device-reachable) addresses."""
        if dd_lo is not None:
            fn = getattr(self.lib, f"ma_{tag}_sum_dd")
            ffi.check(fn(self.handle, addr_of(data), int(n), addr_of(mask), int(mask_bit_offset), int(null_count),
                         addr_of(out_sum), addr_of(dd_lo), addr_of(out_count)))
        else:
            fn = getattr(self.lib, f"ma_{tag}_sum")
            ffi.check(fn(self.handle, addr_of(data), int(n), addr_of(mask), int(mask_bit_offset), int(null_count),
                         addr_of(out_sum), addr_of(out_count)))

    def sum_fused(self, columns) -> None:
        """ONE launch for the sums of up to 4 long 8-byte columns (ma_sum_fused). `columns`: dicts / tuples
        (fmt, data, n, out[, mask, mask_bit_offset, null_count]) with fmt in 'l', 'L', 'g'; `out` is a device-reachable
        address: integers get out[0] = sum, out[1] = count, 'g' gets out[0], out[1] = (hi, lo) bits, out[2] = count."""
        arr = (FusedColumn * len(columns))()
        for i, col in enumerate(columns):
            fmt, data, n, out = col[:4]
            mask = col[4] if len(col) > 4 else None
            arr[i].data = addr_of(data)
            arr[i].n = int(n)
            arr[i].mask_bits = addr_of(mask)
            arr[i].mask_bit_offset = int(col[5]) if len(col) > 5 else 0
            arr[i].null_count = int(col[6]) if len(col) > 6 else -1
            arr[i].format_code = ord(fmt)
            arr[i].reserved = 0
            arr[i].out = addr_of(out)
        ffi.check(self.lib.ma_sum_fused(self.handle, len(columns), C.addressof(arr)))

    def stamp_alloc(self) -> int:
        """Address of 64 bytes of zeroed signal memory a stream can wait on (ma_stamp_alloc)."""
        p = C.c_void_p()
        ffi.check(self.lib.ma_stamp_alloc(self.handle, C.byref(p)))
        return int(p.value)

    def wait_value(self, word: int, value: int) -> None:
        """The context's stream waits until `*word >= value` (ma_ctx_wait_value)."""
        ffi.check(self.lib.ma_ctx_wait_value(self.handle, int(word), int(value)))

    def stamp_free(self, stamp: int) -> None:
        ffi.check(self.lib.ma_stamp_free(self.handle, stamp))

    def prepare_sum_fused(self, columns, stamp: int = 0, early: int = 0):
        """The same call with its argument table built ONCE: returns a zero-argument callable for a stepping host (a Rust
        host builds its ma_fused_column array once too; the Python marshalling is ~30 us per call otherwise)."""
        arr = (FusedColumn * len(columns))()
        for i, col in enumerate(columns):
            fmt, data, n, out = col[:4]
            arr[i].data, arr[i].n, arr[i].out = addr_of(data), int(n), addr_of(out)
            arr[i].mask_bits = addr_of(col[4] if len(col) > 4 else None)
            arr[i].mask_bit_offset = int(col[5]) if len(col) > 5 else 0
            arr[i].null_count = int(col[6]) if len(col) > 6 else -1
            arr[i].format_code, arr[i].reserved = ord(fmt), 0
        handle, k, p = self.handle, len(columns), C.addressof(arr)
        if stamp and early:  # ... and the launch stores it to `early` while it drains (ma_sum_fused_stamped_early)
            fe = self.lib.ma_sum_fused_stamped_early

            def call_stamped_early(value, _keep=arr):
                st = fe(handle, k, p, stamp, value, early)
                if st:
                    ffi.check(st)

            return call_stamped_early
        if stamp:  # call(value): the launch's final thread stores `value` to the stamp behind its results (ma_sum_fused_stamped)
            fs = self.lib.ma_sum_fused_stamped

            def call_stamped(value, _keep=arr):
                st = fs(handle, k, p, stamp, value)
                if st:
                    ffi.check(st)

            return call_stamped
        fn = self.lib.ma_sum_fused

        def call(_keep=arr):
            st = fn(handle, k, p)
            if st:
                ffi.check(st)

        return call

    def sum_columns(self, fmt: str, columns, lens, masks=None, mask_offsets=None):
        """Per-column (sums as float64 array, sums as wrapped int64 array or None for float formats, valid counts)
        for many columns of one type in two launches (ma_sum_columns)."""
        k = len(columns)
        data_arr = (C.c_void_p * k)(*[addr_of(c) or None for c in columns])
        len_arr = (C.c_size_t * k)(*[int(n) for n in lens])
        mask_arr = (C.c_void_p * k)(*[addr_of(m) or None for m in masks]) if masks is not None else None
        off_arr = (C.c_size_t * k)(*[int(o) for o in mask_offsets]) if mask_offsets is not None else None
        f = np.zeros(k, dtype=np.float64)
        i = np.zeros(k, dtype=np.int64)
        c = np.zeros(k, dtype=np.uint64)
        cast = lambda a: C.cast(a, C.c_void_p) if a is not None else None
        ffi.check(self.lib.ma_sum_columns(self.handle, ord(fmt), k, cast(data_arr), cast(len_arr), cast(mask_arr),
                                          cast(off_arr), addr_of(f), addr_of(i), addr_of(c)))
        return f, (i if fmt in "cCsSiIlL" else None), c

    def sum_chunks(self, fmt: str, chunks, lens, masks=None, mask_offsets=None):
        """(sum as float, sum as wrapped int64 or None for float formats, valid count) of ONE column held as a list of
        chunks (ma_sum_chunks): f64 within 1 ULP of the exactly rounded sum."""
        k = len(chunks)
        data_arr = (C.c_void_p * max(k, 1))(*[addr_of(c) or None for c in chunks])
        len_arr = (C.c_size_t * max(k, 1))(*[int(n) for n in lens])
        mask_arr = (C.c_void_p * max(k, 1))(*[addr_of(m) or None for m in masks]) if masks is not None else None
        off_arr = (C.c_size_t * max(k, 1))(*[int(o) for o in mask_offsets]) if mask_offsets is not None else None
        f, i, c = C.c_double(), C.c_int64(), C.c_uint64()
        cast = lambda a: C.cast(a, C.c_void_p) if a is not None else None
        ffi.check(self.lib.ma_sum_chunks(self.handle, ord(fmt), k, cast(data_arr), cast(len_arr), cast(mask_arr), cast(off_arr),
                                         C.addressof(f), C.addressof(i), C.addressof(c)))
        return f.value, (i.value if fmt in "cCsSiIlL" else None), int(c.value)

    def fold_sum_records(self, records, n_records: int, stride_words: int, out4) -> None:
        """Device-side, rank-ordered fold of gathered reduction records (ma_fold_sum_records)."""
        ffi.check(self.lib.ma_fold_sum_records(self.handle, addr_of(records), int(n_records), int(stride_words), addr_of(out4)))

    # -- elementwise arithmetic --------------------------------------------------------------------
    def apply(self, tag: str, lhs, rhs, op: int, out, n_lhs: int, n_rhs: int, mask=None, mask_bit_offset: int = 0,
              out_mask=None) -> None:
        """ma_apply_{int,float}_<tag>(lhs, rhs, op, mask) -> out (+ out_mask). Raises MinarrowHipError with
        .status = MA_ERR_LENGTH_MISMATCH / MA_ERR_DIVIDE_BY_ZERO like the reference's Err / panic."""
        fam = "float" if tag in ("f32", "f64") else "int"
        fn = getattr(self.lib, f"ma_apply_{fam}_{tag}")
        ffi.check(fn(self.handle, addr_of(lhs), int(n_lhs), addr_of(rhs), int(n_rhs), int(op), addr_of(mask),
                     int(mask_bit_offset), addr_of(out), addr_of(out_mask)))

    def apply_scalar(self, tag: str, side: str, arr, n: int, scalar, op: int, out, mask=None,
                     mask_bit_offset: int = 0, out_mask=None) -> None:
        """Fused scalar broadcast: side = "rhs" for array (op) scalar, "lhs" for scalar (op) array."""
        fam = "float" if tag in ("f32", "f64") else "int"
        fn = getattr(self.lib, f"ma_apply_{fam}_{tag}_scalar_{side}")
        sc = float(scalar) if fam == "float" else int(scalar)
        if side == "rhs":
            st = fn(self.handle, addr_of(arr), int(n), sc, int(op), addr_of(mask), int(mask_bit_offset), addr_of(out),
                    addr_of(out_mask))
        else:
            st = fn(self.handle, sc, addr_of(arr), int(n), int(op), addr_of(mask), int(mask_bit_offset), addr_of(out),
                    addr_of(out_mask))
        ffi.check(st)

    def apply_fma(self, tag: str, lhs, rhs, acc, out, n_lhs: int, n_rhs: int, n_acc: int, mask=None,
                  mask_bit_offset: int = 0, out_mask=None) -> None:
        fn = getattr(self.lib, f"ma_apply_fma_{tag}")
        ffi.check(fn(self.handle, addr_of(lhs), int(n_lhs), addr_of(rhs), int(n_rhs), addr_of(acc), int(n_acc),
                     addr_of(mask), int(mask_bit_offset), addr_of(out), addr_of(out_mask)))

    # -- bitmask kernels ---------------------------------------------------------------------------------
    def mask_words_op(self, name: str, lhs, lhs_off: int, rhs, rhs_off: int, n: int, out) -> None:
        """and_masks / or_masks / xor_masks / in_mask / not_in_mask / eq_mask / ne_mask"""
        fn = getattr(self.lib, f"ma_{name}")
        ffi.check(fn(self.handle, addr_of(lhs), int(lhs_off), addr_of(rhs), int(rhs_off), int(n), addr_of(out)))

    def mask_unary_op(self, name: str, src, off: int, n: int, out) -> None:
        """not_mask / bitmask_slice"""
        ffi.check(getattr(self.lib, f"ma_{name}")(self.handle, addr_of(src), int(off), int(n), addr_of(out)))

    def mask_all(self, name: str, a, a_off: int, b, b_off: int, n: int) -> bool:
        """all_eq / all_ne"""
        out = C.c_int32()
        ffi.check(getattr(self.lib, f"ma_{name}")(self.handle, addr_of(a), int(a_off), addr_of(b), int(b_off), int(n),
                                                 C.addressof(out)))
        return bool(out.value)

    def popcount_mask(self, bits, off: int, n: int) -> int:
        out = C.c_uint64()
        ffi.check(self.lib.ma_popcount_mask(self.handle, addr_of(bits), int(off), int(n), C.addressof(out)))
        return int(out.value)

    def all_true_mask(self, bits, n: int) -> bool:
        out = C.c_int32()
        ffi.check(self.lib.ma_all_true_mask(self.handle, addr_of(bits), int(n), C.addressof(out)))
        return bool(out.value)

    def all_false_mask(self, bits, n: int) -> bool:
        out = C.c_int32()
        ffi.check(self.lib.ma_all_false_mask(self.handle, addr_of(bits), int(n), C.addressof(out)))
        return bool(out.value)

    def merge_bitmasks(self, lhs, rhs, n: int, out) -> bool:
        some = C.c_int32()
        ffi.check(self.lib.ma_merge_bitmasks_to_new(self.handle, addr_of(lhs), addr_of(rhs), int(n), addr_of(out),
                                                    C.addressof(some)))
        return bool(some.value)

    def simd_eq_mask(self, tag: str, data, n: int, field_mask: int, target: int, out) -> None:
        ffi.check(getattr(self.lib, f"ma_simd_eq_mask_{tag}")(self.handle, addr_of(data), int(n), int(field_mask),
                                                             int(target), addr_of(out)))

    # -- Arrow C Data Interface --------------------------------------------------------------------------
    def sum_arrow(self, array_ptr: int, schema_ptr: int):
        """(sum as float, sum as wrapped int64, valid_count) of a primitive ArrowArray."""
        f, i, c = C.c_double(), C.c_int64(), C.c_uint64()
        ffi.check(self.lib.ma_sum_arrow(self.handle, int(array_ptr), int(schema_ptr), C.addressof(f), C.addressof(i),
                                        C.addressof(c)))
        return f.value, i.value, int(c.value)

    def mean_arrow(self, array_ptr: int, schema_ptr: int):
        m, c = C.c_double(), C.c_uint64()
        ffi.check(self.lib.ma_mean_arrow(self.handle, int(array_ptr), int(schema_ptr), C.addressof(m), C.addressof(c)))
        return m.value, int(c.value)

    def apply_arrow(self, op: int, lhs_ptrs, rhs_ptrs, out_values, out_validity) -> bool:
        """lhs_ptrs / rhs_ptrs = (ArrowArray*, ArrowSchema*). Returns True when out_validity was written."""
        has = C.c_int32()
        ffi.check(self.lib.ma_apply_arrow(self.handle, int(op), int(lhs_ptrs[0]), int(lhs_ptrs[1]), int(rhs_ptrs[0]),
                                          int(rhs_ptrs[1]), addr_of(out_values), addr_of(out_validity),
                                          C.addressof(has)))
        return bool(has.value)

    def apply_arrow_export(self, op: int, lhs_ptrs, rhs_ptrs, name: Optional[str] = None):
        """lhs (op) rhs returned as a library-owned Arrow C Data pair (`arrow_c.Owned`)."""
        from .arrow_c import Owned

        out = Owned()
        ffi.check(self.lib.ma_apply_arrow_export(self.handle, int(op), int(lhs_ptrs[0]), int(lhs_ptrs[1]),
                                                 int(rhs_ptrs[0]), int(rhs_ptrs[1]),
                                                 name.encode() if name is not None else None, out.array_ptr,
                                                 out.schema_ptr))
        return out

    def apply_arrow_batch_export(self, op: int, lhs_ptrs, rhs_ptrs):
        """Table (op) Table over two record batches (struct arrays); returns `arrow_c.Owned` holding a struct array."""
        from .arrow_c import Owned

        out = Owned()
        ffi.check(self.lib.ma_apply_arrow_batch_export(self.handle, int(op), int(lhs_ptrs[0]), int(lhs_ptrs[1]),
                                                       int(rhs_ptrs[0]), int(rhs_ptrs[1]), out.array_ptr,
                                                       out.schema_ptr))
        return out

    # -- consolidation of a chunked column (SuperTable / SuperArray) ------------------------------------
    def consolidate_column(self, elem_size: int, chunks, lens, out_data, masks=None, mask_offsets=None,
                           out_mask=None) -> bool:
        """chunks / masks: sequences of buffers (mask entries may be None). Returns True when out_mask was written."""
        k = len(chunks)
        data_arr = (C.c_void_p * k)(*[addr_of(c) or None for c in chunks])
        len_arr = (C.c_size_t * k)(*[int(n) for n in lens])
        mask_arr = (C.c_void_p * k)(*[addr_of(m) or None for m in masks]) if masks is not None else None
        off_arr = (C.c_size_t * k)(*[int(o) for o in mask_offsets]) if mask_offsets is not None else None
        has = C.c_int32()
        ffi.check(self.lib.ma_consolidate_column(
            self.handle, int(elem_size), k, C.cast(data_arr, C.c_void_p), C.cast(len_arr, C.c_void_p),
            C.cast(mask_arr, C.c_void_p) if mask_arr is not None else None,
            C.cast(off_arr, C.c_void_p) if off_arr is not None else None, addr_of(out_data), addr_of(out_mask),
            C.addressof(has)))
        return bool(has.value)

    def consolidate_table_arena(self, elem_sizes, batch_rows, cells, arena, arena_bytes: int, cell_masks=None,
                                cell_mask_offsets=None):
        """Whole-table consolidation into one arena (ma_consolidate_table_arena; src/structs/arena.rs:1187-1340).
        cells[c][b] = buffer of column c, batch b; cell_masks[c][b] = validity buffer or None. Returns
        (data offsets, mask offsets with None for columns without nulls, used bytes)."""
        n_cols, n_batches = len(elem_sizes), len(batch_rows)
        k = n_cols * n_batches
        flat = [cells[c][b] for c in range(n_cols) for b in range(n_batches)]
        data_arr = (C.c_void_p * k)(*[addr_of(x) or None for x in flat])
        mask_arr = off_arr = None
        if cell_masks is not None:
            mask_arr = (C.c_void_p * k)(*[addr_of(cell_masks[c][b]) or None for c in range(n_cols) for b in range(n_batches)])
        if cell_mask_offsets is not None:
            off_arr = (C.c_size_t * k)(*[int(cell_mask_offsets[c][b]) for c in range(n_cols) for b in range(n_batches)])
        es = (C.c_size_t * n_cols)(*[int(e) for e in elem_sizes])
        br = (C.c_size_t * n_batches)(*[int(r) for r in batch_rows])
        d_off, m_off, used = (C.c_size_t * n_cols)(), (C.c_size_t * n_cols)(), C.c_size_t()
        cast = lambda a: C.cast(a, C.c_void_p) if a is not None else None
        ffi.check(self.lib.ma_consolidate_table_arena(self.handle, n_cols, n_batches, cast(es), cast(br), cast(data_arr),
                                                      cast(mask_arr), cast(off_arr), addr_of(arena), int(arena_bytes),
                                                      cast(d_off), cast(m_off), C.addressof(used)))
        none = (1 << 64) - 1
        return list(d_off), [None if m == none else int(m) for m in m_off], int(used.value)

    def consolidate_boolean_column(self, chunks, out_bits, masks=None, out_mask=None) -> bool:
        """chunks: [(bits buffer, bit offset, len)]; masks: [(bits buffer, bit offset) or None]. Returns True when
        out_mask was written (BooleanArray::append_range, src/structs/variants/boolean.rs:627-653)."""
        k = len(chunks)
        bits_arr = (C.c_void_p * k)(*[addr_of(c[0]) or None for c in chunks])
        off_arr = (C.c_size_t * k)(*[int(c[1]) for c in chunks])
        len_arr = (C.c_size_t * k)(*[int(c[2]) for c in chunks])
        mask_arr = moff_arr = None
        if masks is not None:
            mask_arr = (C.c_void_p * k)(*[(addr_of(m[0]) if m is not None else None) for m in masks])
            moff_arr = (C.c_size_t * k)(*[(int(m[1]) if m is not None else 0) for m in masks])
        has = C.c_int32()
        cast = lambda a: C.cast(a, C.c_void_p) if a is not None else None
        ffi.check(self.lib.ma_consolidate_boolean_column(self.handle, k, cast(bits_arr), cast(off_arr), cast(len_arr),
                                                         cast(mask_arr), cast(moff_arr), addr_of(out_bits),
                                                         addr_of(out_mask), C.addressof(has)))
        return bool(has.value)

    def apply_datetime(self, tag: str, lhs, lhs_off: int, lhs_len: int, lhs_mask, rhs, rhs_off: int, rhs_len: int,
                       rhs_mask, op: int, out, out_mask) -> bool:
        """apply_datetime_<tag>((lhs, off, len), (rhs, off, len), op). Returns True when out_mask was written."""
        has = C.c_int32()
        fn = getattr(self.lib, f"ma_apply_datetime_{tag}")
        ffi.check(fn(self.handle, addr_of(lhs), int(lhs_off), int(lhs_len), addr_of(lhs_mask), addr_of(rhs), int(rhs_off),
                     int(rhs_len), addr_of(rhs_mask), int(op), addr_of(out), addr_of(out_mask), C.addressof(has)))
        return bool(has.value)

    def apply_promote(self, ltag: str, rtag: str, lhs, rhs, op: int, out, n_lhs: int, n_rhs: int, mask=None,
                      mask_bit_offset: int = 0, out_mask=None) -> None:
        """(Int32, Float64/Float32) or (Float64/Float32, Int32): promotion fused (routing/arithmetic.rs:342-373)."""
        fn = getattr(self.lib, f"ma_apply_promote_{ltag}_{rtag}")
        ffi.check(fn(self.handle, addr_of(lhs), int(n_lhs), addr_of(rhs), int(n_rhs), int(op), addr_of(mask),
                     int(mask_bit_offset), addr_of(out), addr_of(out_mask)))

    def route_super_array_broadcast(self, fmt: str, op: int, lhs_chunks, rhs_chunks, lens_l, lens_r, out_chunks,
                                    lhs_masks=None, rhs_masks=None, out_masks=None, override=None):
        """SuperArray (op) SuperArray, chunk by chunk (src/kernels/broadcast/super_array.rs:180-251).
        Returns the per-chunk "has validity" flags."""
        k = len(lhs_chunks)

        def table(items):
            return C.cast((C.c_void_p * k)(*[addr_of(x) or None for x in items]), C.c_void_p) if items is not None else None

        ll = (C.c_size_t * k)(*[int(n) for n in lens_l])
        lr = (C.c_size_t * k)(*[int(n) for n in lens_r])
        has = (C.c_int32 * k)()
        ffi.check(self.lib.ma_route_super_array_broadcast(
            self.handle, ord(fmt), int(op), k, table(lhs_chunks), C.cast(ll, C.c_void_p), table(lhs_masks),
            table(rhs_chunks), C.cast(lr, C.c_void_p), table(rhs_masks), addr_of(override), table(out_chunks),
            table(out_masks), C.cast(has, C.c_void_p)))
        return [bool(x) for x in has]

    def broadcast_super_array_scalar(self, fmt: str, op: int, scalar, chunks, lens, out_chunks, masks=None, out_masks=None,
                                     scalar_is_lhs: bool = False):
        """SuperArray (op) Scalar, or Scalar (op) SuperArray, all chunks in one launch (super_array.rs:87-116,
        scalar.rs:214-243). `scalar`: a numpy scalar / 1-element array of the chunks' type. Returns the "has validity" flags."""
        k = len(chunks)

        def table(items):
            return C.cast((C.c_void_p * k)(*[addr_of(x) or None for x in items]), C.c_void_p) if items is not None else None

        sc = np.ascontiguousarray(np.asarray(scalar).reshape(1))
        ll = (C.c_size_t * k)(*[int(n) for n in lens])
        has = (C.c_int32 * k)()
        ffi.check(self.lib.ma_broadcast_super_array_scalar(
            self.handle, ord(fmt), int(op), 1 if scalar_is_lhs else 0, addr_of(sc), k, table(chunks), C.cast(ll, C.c_void_p),
            table(masks), table(out_chunks), table(out_masks), C.cast(has, C.c_void_p)))
        return [bool(x) for x in has]

    def apply_arrow_stream_export(self, op: int, lhs_stream_ptr: int, rhs_stream_ptr: int, out_stream_ptr: int) -> None:
        """SuperTable (op) SuperTable as a stream operator: moves both input ArrowArrayStreams, fills *out_stream."""
        ffi.check(self.lib.ma_apply_arrow_stream_export(self.handle, int(op), int(lhs_stream_ptr), int(rhs_stream_ptr),
                                                        int(out_stream_ptr)))

    def sum_arrow_stream(self, stream_ptr: int, column: int = 0):
        """(sum as float, sum as wrapped int64, valid_count, rows, batches) over every batch of an ArrowArrayStream."""
        f, i = C.c_double(), C.c_int64()
        c, r, b = C.c_uint64(), C.c_uint64(), C.c_uint64()
        ffi.check(self.lib.ma_sum_arrow_stream(self.handle, int(stream_ptr), int(column), C.addressof(f), C.addressof(i),
                                               C.addressof(c), C.addressof(r), C.addressof(b)))
        return f.value, i.value, int(c.value), int(r.value), int(b.value)


class SelftestReport(C.Structure):
    """ma_selftest_report (include/minarrow_hip.h)."""
    _fields_ = [("struct_bytes", C.c_uint32), ("n_members", C.c_int32), ("n_devices", C.c_int32), ("exchange_kind", C.c_int32),
                ("rccl_ranks", C.c_int32), ("forms_tried", C.c_uint32), ("forms_ok", C.c_uint32), ("peer_pairs", C.c_int32),
                ("peer_pairs_ok", C.c_int32), ("stamp_waits", C.c_int32), ("stamp_waits_ok", C.c_int32), ("failed_form", C.c_int32),
                ("failed_member", C.c_int32), ("timed_out", C.c_int32), ("form_us", C.c_double * 8), ("peer_us_max", C.c_double),
                ("stamp_us_max", C.c_double), ("text", C.c_char * 1024)]

    FORM_NAMES = ("in-stream/threads", "in-stream/caller", "overlap-event/threads", "overlap-event/caller",
                  "overlap-stamp/threads", "overlap-stamp/caller", "host-fold")

    def as_dict(self, status: int = 0) -> dict:
        forms = {name: {"ok": bool(self.forms_ok >> b & 1), "us": round(self.form_us[b], 1)}
                 for b, name in enumerate(self.FORM_NAMES) if self.forms_tried >> b & 1}
        return {"ok": status == 0, "text": self.text.decode(errors="replace"), "members": self.n_members, "devices": self.n_devices,
                "exchange": "rccl" if self.exchange_kind == 1 else "host", "rccl_ranks": self.rccl_ranks, "forms": forms,
                "peer_pairs": self.peer_pairs, "peer_pairs_ok": self.peer_pairs_ok, "peer_us_max": round(self.peer_us_max, 1),
                "stamp_waits": self.stamp_waits, "stamp_waits_ok": self.stamp_waits_ok, "stamp_us_max": round(self.stamp_us_max, 1),
                "failed_form": self.FORM_NAMES[self.failed_form] if 0 <= self.failed_form < len(self.FORM_NAMES) else None,
                "failed_member": self.failed_member if self.failed_member >= 0 else None, "timed_out": bool(self.timed_out)}


FORM_BITS = 16 | 32 | 128 | 256 | 16384 | 65536  # ma_common.hpp kFormBits: the ctx variant bits every build of the library takes


def tuning_build() -> bool:
    """True when the loaded library is the tuning build (MINARROW_HIP_LIB=build/tuning/libminarrow_hip.so: make -C minarrow_amd/csrc
    TUNING=1), whose ma_ctx_set_variant takes every bit."""
    return "tuning" in str(ffi.LIB_PATH)


def live_variants(variants):
    """Those of `variants` this build's ma_ctx_set_variant accepts: form bits always, tuning bits on the tuning build only."""
    return [v for v in variants if tuning_build() or (v & ~FORM_BITS) == 0]


SELFTEST_EXCHANGE, SELFTEST_EXCHANGE_ALL_FORMS, SELFTEST_PEER_COPIES, SELFTEST_STAMPS = 1, 2, 4, 8
SELFTEST_OVERLAP_EVENT, SELFTEST_OVERLAP_STAMP = 16, 32  # Comm.selftest only
GROUP_FLAGS = {"host": 0, "rccl": 1, "rccl-or-host": 3, "rccl-overlap": 1 | 8, "rccl-overlap-or-host": 3 | 8,
               "rccl-overlap-lanes": 1 | 8 | 16, "rccl-overlap-lanes-or-host": 3 | 8 | 16}


class ScanLanes:
    """Back-to-back fused sums on one GPU as a pipeline (ma_scan_lanes_*): consecutive scans on two streams of the context's
    device, each started when the one before it has begun to drain — the reference's hot loop of sums
    (benches/hotloop_benchmark_avg_std.rs:48-62: ITERATIONS passes, an i64 and an f64 sum each; the pass itself: hotloop_benchmark_std.rs:109-127) without a launch's fixed cost between the scans."""

    def __init__(self, ctx: "Context"):
        self.ctx, self.lib = ctx, ctx.lib
        h = C.c_void_p()
        ffi.check(self.lib.ma_scan_lanes_create(ctx.handle, C.byref(h)))
        self.handle = h.value
        if not hasattr(ctx, "_scan_lanes"):
            ctx._scan_lanes = []
        ctx._scan_lanes.append(self)

    def prepare_sum_fused(self, columns):
        """A zero-argument callable that enqueues one fused scan of `columns` ((format, data, n, out[, mask, bit offset[,
        null count]]) each, as Context.sum_fused) on the lane whose turn it is; the argument table is built once."""
        arr = (FusedColumn * len(columns))()
        for i, col in enumerate(columns):
            fmt, data, n, out = col[:4]
            arr[i].data, arr[i].n, arr[i].out = addr_of(data), int(n), addr_of(out)
            arr[i].mask_bits = addr_of(col[4] if len(col) > 4 else None)
            arr[i].mask_bit_offset = int(col[5]) if len(col) > 5 else 0
            arr[i].null_count = int(col[6]) if len(col) > 6 else -1
            arr[i].format_code, arr[i].reserved = ord(fmt), 0
        fn, handle, k, p = self.lib.ma_scan_lanes_sum_fused, self.handle, len(columns), C.addressof(arr)

        def call(_keep=arr):
            st = fn(handle, k, p)
            if st:
                ffi.check(st)

        return call

    def sum_fused(self, columns) -> None:
        self.prepare_sum_fused(columns)()

    def prepare_sum(self, fmt: str, data, n: int, out_sum, out_count=None, out_lo=None, mask=None, mask_bit_offset: int = 0,
                    null_count: int = -1):
        """A zero-argument callable that enqueues ONE column of any numeric type (Arrow format character) on the lane whose
        turn it is (ma_scan_lanes_sum): the single-column kernels of ma_<t>_sum, pipelined."""
        fn, handle = self.lib.ma_scan_lanes_sum, self.handle
        args = (ord(fmt), addr_of(data), int(n), addr_of(mask), int(mask_bit_offset), int(null_count), addr_of(out_sum),
                addr_of(out_lo), addr_of(out_count))

        def call():
            st = fn(handle, *args)
            if st:
                ffi.check(st)

        return call

    def sum(self, fmt: str, data, n: int, out_sum, out_count=None, out_lo=None, mask=None, mask_bit_offset: int = 0,
            null_count: int = -1) -> None:
        self.prepare_sum(fmt, data, n, out_sum, out_count, out_lo, mask, mask_bit_offset, null_count)()

    def join(self) -> None:
        ffi.check(self.lib.ma_scan_lanes_join(self.handle))

    def synchronize(self) -> None:
        ffi.check(self.lib.ma_scan_lanes_synchronize(self.handle))

    def synchronize_for(self, timeout_ms: float) -> None:
        """ma_scan_lanes_synchronize_for: past the deadline every gate of the pipeline is released, the pipeline is broken
        (is_broken; destroy it) and MinarrowHipError(MA_ERR_DEVICE) names the lane and the sequence that was waited for."""
        ffi.check(self.lib.ma_scan_lanes_synchronize_for(self.handle, float(timeout_ms)))

    @property
    def is_broken(self) -> int:
        """0 healthy, 1 broken with both streams empty, 2 broken with a stream still busy."""
        return int(self.lib.ma_scan_lanes_is_broken(self.handle))

    def test_hold_next_scan(self) -> None:
        """TESTING ONLY (include/minarrow_hip_testing.h): the next scan sits behind a gate that never opens."""
        ffi.check(self.lib.ma_scan_lanes_test_hold_next_scan(self.handle))

    @property
    def scans(self) -> int:
        return int(self.lib.ma_scan_lanes_scans(self.handle))

    def close(self) -> None:
        if self.handle:
            self.lib.ma_scan_lanes_destroy(self.handle)
            self.handle = None
            if self in getattr(self.ctx, "_scan_lanes", ()):
                self.ctx._scan_lanes.remove(self)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Group:
    """One process driving several GPUs (ma_group_*): member i scans chunk i on device i, one exchange ends the
    reduction. exchange = "rccl" (ncclCommInitAll + grouped all-gather + device fold), "rccl-or-host", or "host"."""

    MAX_COLUMNS = 16

    def __init__(self, devices, exchange: str = "host", issue: str = "threads"):
        self.lib = ffi.load_library()
        flags = GROUP_FLAGS[exchange] | {"threads": 0, "caller": 4}[issue]
        n = len(devices)
        devs = (C.c_int32 * n)(*[int(d) for d in devices])
        h = C.c_void_p()
        ffi.check(self.lib.ma_group_create_ex(C.cast(devs, C.c_void_p), n, flags, C.byref(h)))
        self.handle = h.value
        self.size = n
        self.devices = [int(d) for d in devices]
        self._members = []

    @property
    def exchange_kind(self) -> str:
        return "rccl" if self.lib.ma_group_exchange_kind(self.handle) == 1 else "host"

    @property
    def exchange_note(self) -> str:
        s = self.lib.ma_group_exchange_note(self.handle)
        return s.decode() if s else ""

    @property
    def issue_kind(self) -> str:
        """"threads": every member's launches are enqueued by its own issue thread; "caller": by the calling thread."""
        return "threads" if self.lib.ma_group_issue_kind(self.handle) == 1 else "caller"

    def test_set_member_device(self, member: int, hip_device: int, peer_capable: bool = True) -> None:
        """TESTING ONLY (ma_group_test_set_member_device): the checks treat `member` as living on `hip_device`."""
        ffi.check(self.lib.ma_group_test_set_member_device(self.handle, int(member), int(hip_device), 1 if peer_capable else 0))

    def peer_access(self, from_member: int, to_member: int) -> bool:
        return self.lib.ma_group_peer_access(self.handle, int(from_member), int(to_member)) == 1

    def member_ctx(self, i: int) -> "Context":
        """A non-owning Context over member i's ma_ctx (allocate / fill that device's chunk through it)."""
        c = Context.__new__(Context)
        c.lib = self.lib
        c.handle = self.lib.ma_group_ctx(self.handle, int(i))
        c.device = self.devices[i]
        c.close = lambda: None  # the group owns it
        c._buffers = weakref.WeakSet()
        self._members.append(c)
        return c

    def _tables(self, chunks, lens, masks, mask_offsets):
        k = self.size
        assert len(chunks) == k and len(lens) == k
        data = (C.c_void_p * k)(*[addr_of(c) or None for c in chunks])
        ln = (C.c_size_t * k)(*[int(n) for n in lens])
        m = (C.c_void_p * k)(*[addr_of(x) or None for x in masks]) if masks is not None else None
        o = (C.c_size_t * k)(*[int(x) for x in mask_offsets]) if mask_offsets is not None else None
        cast = lambda a: C.cast(a, C.c_void_p) if a is not None else None
        return cast(data), cast(ln), cast(m), cast(o), (data, ln, m, o)

    def enqueue_sum(self, tag: str, column: int, chunks, lens, masks=None, mask_offsets=None) -> None:
        d, l, m, o, keep = self._tables(chunks, lens, masks, mask_offsets)
        ffi.check(getattr(self.lib, f"ma_group_enqueue_sum_{tag}")(self.handle, int(column), d, l, m, o))

    def enqueue_sum_table(self, columns) -> None:
        """ONE fused launch per member for several partitioned 8-byte columns (ma_group_enqueue_sum_table).
        `columns`: tuples (fmt, record_slot, chunks, lens[, masks, mask_offsets]) with fmt in 'l', 'L', 'g'."""
        k = len(columns)
        n = self.size
        slots = (C.c_int32 * k)(*[int(c[1]) for c in columns])
        fmts = (C.c_int32 * k)(*[ord(c[0]) for c in columns])
        keep = []
        data_t, lens_t, masks_t, offs_t = ((C.c_void_p * k)() for _ in range(4))
        any_mask = False
        for j, c in enumerate(columns):
            chunks, lens = c[2], c[3]
            masks = c[4] if len(c) > 4 else None
            offs = c[5] if len(c) > 5 else None
            d = (C.c_void_p * n)(*[addr_of(x) for x in chunks])
            ln = (C.c_size_t * n)(*[int(x) for x in lens])
            keep += [d, ln]
            data_t[j], lens_t[j] = C.addressof(d), C.addressof(ln)
            if masks is not None:
                m = (C.c_void_p * n)(*[addr_of(x) for x in masks])
                o = (C.c_size_t * n)(*[int(x) for x in (offs or [0] * n)])
                keep += [m, o]
                masks_t[j], offs_t[j] = C.addressof(m), C.addressof(o)
                any_mask = True
        ffi.check(self.lib.ma_group_enqueue_sum_table(self.handle, k, slots, fmts, data_t, lens_t, masks_t if any_mask else None,
                                                      offs_t if any_mask else None))

    def prepare_sum_table(self, columns):
        """enqueue_sum_table with its pointer tables built ONCE: a zero-argument callable for a stepping host."""
        k = len(columns)
        n = self.size
        slots = (C.c_int32 * k)(*[int(c[1]) for c in columns])
        fmts = (C.c_int32 * k)(*[ord(c[0]) for c in columns])
        keep = [slots, fmts]
        data_t, lens_t, masks_t, offs_t = ((C.c_void_p * k)() for _ in range(4))
        any_mask = False
        for j, c in enumerate(columns):
            masks = c[4] if len(c) > 4 else None
            offs = c[5] if len(c) > 5 else None
            d = (C.c_void_p * n)(*[addr_of(x) for x in c[2]])
            ln = (C.c_size_t * n)(*[int(x) for x in c[3]])
            keep += [d, ln]
            data_t[j], lens_t[j] = C.addressof(d), C.addressof(ln)
            if masks is not None:
                m = (C.c_void_p * n)(*[addr_of(x) for x in masks])
                o = (C.c_size_t * n)(*[int(x) for x in (offs or [0] * n)])
                keep += [m, o]
                masks_t[j], offs_t[j] = C.addressof(m), C.addressof(o)
                any_mask = True
        keep += [data_t, lens_t, masks_t, offs_t]
        fn, handle = self.lib.ma_group_enqueue_sum_table, self.handle
        args = (handle, k, slots, fmts, data_t, lens_t, masks_t if any_mask else None, offs_t if any_mask else None)

        def call(_keep=keep):
            st = fn(*args)
            if st:
                ffi.check(st)

        return call

    def exchange_stats(self):
        """{all_gather_us, fold_us, samples, rccl_ranks} of the exchanges sampled since the last call (ma_group_exchange_stats)."""
        g, f, k, r = C.c_double(), C.c_double(), C.c_int32(), C.c_int32()
        ffi.check(self.lib.ma_group_exchange_stats(self.handle, C.byref(g), C.byref(f), C.byref(k), C.byref(r)))
        return {"all_gather_us": g.value, "fold_us": f.value, "samples": int(k.value), "rccl_ranks": int(r.value)}

    def enqueue_sum_chunks(self, fmt: str, column: int, chunks, lens, masks=None, mask_offsets=None) -> None:
        """ONE column held as many chunks, chunk i on member i % size (ma_group_enqueue_sum_chunks). Enqueues only:
        exchange(), synchronize(), then result(column)."""
        k = len(chunks)
        data_arr = (C.c_void_p * max(k, 1))(*[addr_of(c) or None for c in chunks])
        len_arr = (C.c_size_t * max(k, 1))(*[int(n) for n in lens])
        mask_arr = (C.c_void_p * max(k, 1))(*[addr_of(m) or None for m in masks]) if masks is not None else None
        off_arr = (C.c_size_t * max(k, 1))(*[int(o) for o in mask_offsets]) if mask_offsets is not None else None
        cast = lambda a: C.cast(a, C.c_void_p) if a is not None else None
        ffi.check(self.lib.ma_group_enqueue_sum_chunks(self.handle, int(column), ord(fmt), k, cast(data_arr), cast(len_arr),
                                                       cast(mask_arr), cast(off_arr)))

    def route_super_array_broadcast(self, fmt: str, op: int, lhs_chunks, rhs_chunks, lens_l, lens_r, out_chunks,
                                    lhs_masks=None, rhs_masks=None, out_masks=None, member_overrides=None):
        """SuperArray (op) SuperArray over the group's GPUs (ma_group_route_super_array_broadcast): chunk pair i runs on
        member i % size, where its buffers live. Enqueues only — call synchronize(). Returns the "has validity" flags."""
        k = len(lhs_chunks)

        def table(items, n=k):
            return C.cast((C.c_void_p * n)(*[addr_of(x) or None for x in items]), C.c_void_p) if items is not None else None

        ll = (C.c_size_t * k)(*[int(n) for n in lens_l])
        lr = (C.c_size_t * k)(*[int(n) for n in lens_r])
        has = (C.c_int32 * k)()
        ffi.check(self.lib.ma_group_route_super_array_broadcast(
            self.handle, ord(fmt), int(op), k, table(lhs_chunks), C.cast(ll, C.c_void_p), table(lhs_masks),
            table(rhs_chunks), C.cast(lr, C.c_void_p), table(rhs_masks), table(member_overrides, self.size),
            table(out_chunks), table(out_masks), C.cast(has, C.c_void_p)))
        return [bool(x) for x in has]

    def consolidate_column(self, dest_member: int, elem_size: int, chunks, lens, out_data, masks=None, mask_offsets=None,
                           out_mask=None) -> bool:
        """SuperTable::consolidate for a column sharded over the group (ma_group_consolidate_column): chunk i lives on
        member i % size, the result on member dest_member. Enqueues only. Returns True when out_mask will be written."""
        k = len(chunks)
        data_arr = (C.c_void_p * k)(*[addr_of(c) or None for c in chunks])
        len_arr = (C.c_size_t * k)(*[int(n) for n in lens])
        mask_arr = (C.c_void_p * k)(*[addr_of(m) or None for m in masks]) if masks is not None else None
        off_arr = (C.c_size_t * k)(*[int(o) for o in mask_offsets]) if mask_offsets is not None else None
        has = C.c_int32()
        ffi.check(self.lib.ma_group_consolidate_column(
            self.handle, int(dest_member), int(elem_size), k, C.cast(data_arr, C.c_void_p), C.cast(len_arr, C.c_void_p),
            C.cast(mask_arr, C.c_void_p) if mask_arr is not None else None,
            C.cast(off_arr, C.c_void_p) if off_arr is not None else None, addr_of(out_data), addr_of(out_mask),
            C.addressof(has)))
        return bool(has.value)

    def exchange(self) -> None:
        ffi.check(self.lib.ma_group_exchange(self.handle))

    def synchronize(self) -> None:
        ffi.check(self.lib.ma_group_synchronize(self.handle))

    def synchronize_for(self, timeout_ms: float) -> None:
        """ma_group_synchronize_for: raises MinarrowHipError(MA_ERR_DEVICE) naming the pending members past the deadline;
        the group is broken then (is_broken) until rebuild_exchange."""
        ffi.check(self.lib.ma_group_synchronize_for(self.handle, float(timeout_ms)))

    @property
    def is_broken(self) -> int:
        """0 healthy, 1 broken with idle streams (rebuild_exchange works), 2 broken with a stream still busy."""
        return int(self.lib.ma_group_is_broken(self.handle))

    @property
    def flags(self) -> int:
        return int(self.lib.ma_group_flags(self.handle))

    @property
    def overlapped(self) -> bool:
        return bool(self.flags & 8)

    @property
    def scan_lanes(self) -> bool:
        """Two scan streams per member, consecutive stamped steps gated on each other's early stamp (MA_GROUP_SCAN_LANES)."""
        return bool(self.flags & 16)

    def mark_next_scan(self, from_index: int, to_index: int) -> None:
        ffi.check(self.lib.ma_group_mark_next_scan(self.handle, int(from_index), int(to_index)))

    def mark_elapsed_ms(self, member: int, from_index: int, to_index: int) -> float:
        ms = C.c_float()
        ffi.check(self.lib.ma_group_mark_elapsed_ms(self.handle, int(member), int(from_index), int(to_index), C.addressof(ms)))
        return float(ms.value)

    def set_scan_lanes(self, on) -> None:
        """False / True, or 2 = on, with fresh second contexts (new streams: the runtime maps them onto hardware queues anew)."""
        ffi.check(self.lib.ma_group_set_scan_lanes(self.handle, int(on)))

    def join_lanes(self) -> None:
        ffi.check(self.lib.ma_group_join_lanes(self.handle))

    @property
    def handoff(self) -> Optional[str]:
        """"stamp" / "event": what an overlapped, stamped step's exchange stream waits on; None when the group does not overlap."""
        return {0: "stamp", 1: "event"}.get(int(self.lib.ma_group_handoff(self.handle)))

    def set_handoff(self, kind: str) -> None:
        ffi.check(self.lib.ma_group_set_handoff(self.handle, {"stamp": 0, "event": 1}[kind]))

    def rebuild_exchange(self, exchange: str, issue: str = "threads") -> None:
        """A fresh exchange for the same members (ma_group_rebuild_exchange): the member contexts and their buffers stay."""
        ffi.check(self.lib.ma_group_rebuild_exchange(self.handle, GROUP_FLAGS[exchange] | {"threads": 0, "caller": 4}[issue]))

    def selftest(self, timeout_ms: float, what: int = 0, raise_on_failure: bool = False) -> dict:
        """ma_group_selftest: the report as a dict ({"ok", "text", "forms", ...})."""
        rep = SelftestReport()
        st = self.lib.ma_group_selftest(self.handle, int(what), float(timeout_ms), C.addressof(rep))
        out = rep.as_dict(st)
        if st != 0:
            msg = self.lib.ma_last_error_string()
            out["error"] = msg.decode() if msg else ""
            if raise_on_failure:
                raise ffi.MinarrowHipError(st, out["error"])
        return out

    def test_fail_next_exchange(self, member: int) -> None:
        ffi.check(self.lib.ma_group_test_fail_next_exchange(self.handle, int(member)))

    def test_stall_next_exchange(self, member: int) -> None:
        ffi.check(self.lib.ma_group_test_stall_next_exchange(self.handle, int(member)))

    def test_corrupt_next_exchange(self, member: int) -> None:
        ffi.check(self.lib.ma_group_test_corrupt_next_exchange(self.handle, int(member)))

    def result(self, column: int = 0, member: int = 0):
        """(int sum, int count, f64 sum, f64 count) of `column` as GPU `member` holds them (after synchronize)."""
        i, ic, f, fc = C.c_int64(), C.c_uint64(), C.c_double(), C.c_uint64()
        ffi.check(self.lib.ma_group_member_result(self.handle, int(member), int(column), C.addressof(i), C.addressof(ic),
                                                  C.addressof(f), C.addressof(fc)))
        return int(i.value), int(ic.value), float(f.value), int(fc.value)

    def sum(self, tag: str, chunks, lens, masks=None, mask_offsets=None):
        """Synchronous one-call form: (sum, valid count)."""
        d, l, m, o, keep = self._tables(chunks, lens, masks, mask_offsets)
        cnt = C.c_uint64()
        out = C.c_int64() if tag == "i64" else C.c_double()
        ffi.check(getattr(self.lib, f"ma_group_sum_{tag}")(self.handle, d, l, m, o, C.addressof(out), C.addressof(cnt)))
        return out.value, int(cnt.value)

    def close(self) -> None:
        if self.handle:
            # A bounded drain first: freeing device memory waits for the device, and a stream that is still held (a collective
            # whose peer never came; a stall hook) would keep that wait for good. Past the deadline the library aborts the
            # communicators and releases what it can — the group is going away anyway.
            if self.lib.ma_group_is_broken(self.handle) == 0:
                self.lib.ma_group_synchronize_for(self.handle, 2000.0)
            for c in self._members:  # buffers still alive on a member's device go first; the views die with the group
                for buf in list(c._buffers):
                    buf.free()
                c.handle = None
            self.lib.ma_group_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Comm:
    """One rank of a multi-process RCCL communicator bound to a Context (ma_comm_*)."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * Comm.ID_BYTES)()
        ffi.check(ffi.load_library().ma_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    def __init__(self, ctx: "Context", unique_id: bytes, rank: int, n_ranks: int):
        assert len(unique_id) == self.ID_BYTES
        self.ctx, self.lib = ctx, ctx.lib
        buf = (C.c_uint8 * self.ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        ffi.check(self.lib.ma_comm_create(ctx.handle, C.cast(buf, C.c_void_p), int(rank), int(n_ranks), C.byref(h)))
        self.handle = h.value
        self.rank, self.size = int(rank), int(n_ranks)

    def all_gather(self, send, recv, bytes_per_rank: int) -> None:
        ffi.check(self.lib.ma_comm_all_gather(self.handle, addr_of(send), addr_of(recv), int(bytes_per_rank)))

    def all_reduce_sum_i64(self, send, recv, count: int) -> None:
        ffi.check(self.lib.ma_comm_all_reduce_sum_i64(self.handle, addr_of(send), addr_of(recv), int(count)))

    def sum_exchange(self, local_records, slots_per_rank: int, n_columns: int, gathered, out_finals) -> None:
        ffi.check(self.lib.ma_comm_sum_exchange(self.handle, addr_of(local_records), int(slots_per_rank), int(n_columns),
                                                addr_of(gathered), addr_of(out_finals)))

    def sum_exchange_overlapped(self, slot: int, local_records, slots_per_rank: int, n_columns: int, gathered, out_finals) -> None:
        """The exchange of record set `slot` (0 / 1) on the communicator's own stream; the context's stream goes on at once."""
        ffi.check(self.lib.ma_comm_sum_exchange_overlapped(self.handle, int(slot), addr_of(local_records), int(slots_per_rank),
                                                           int(n_columns), addr_of(gathered), addr_of(out_finals)))

    def exchange_stats(self):
        g, f, k, r = C.c_double(), C.c_double(), C.c_int32(), C.c_int32()
        ffi.check(self.lib.ma_comm_exchange_stats(self.handle, C.byref(g), C.byref(f), C.byref(k), C.byref(r)))
        return {"all_gather_us": g.value, "fold_us": f.value, "samples": int(k.value), "rccl_ranks": int(r.value)}

    def sum_exchange_overlapped_on_stamp(self, slot: int, stamp: int, value: int, local_records, slots_per_rank: int, n_columns: int,
                                         gathered, out_finals) -> None:
        """As sum_exchange_overlapped, but the exchange stream waits for `*stamp >= value` instead of an event on the scan stream."""
        ffi.check(self.lib.ma_comm_sum_exchange_overlapped_on_stamp(self.handle, int(slot), stamp, int(value), addr_of(local_records),
                                                                   int(slots_per_rank), int(n_columns), addr_of(gathered), addr_of(out_finals)))

    def slot_wait(self, slot: int, ctx: Optional["Context"] = None) -> None:
        """Puts the context's stream (or `ctx`'s: a second scan context of the same device) behind the last overlapped exchange
        of record set `slot`."""
        if ctx is None:
            ffi.check(self.lib.ma_comm_slot_wait(self.handle, int(slot)))
        else:
            ffi.check(self.lib.ma_comm_slot_wait_on(self.handle, int(slot), ctx.handle))

    def synchronize(self) -> None:
        ffi.check(self.lib.ma_comm_synchronize(self.handle))

    def synchronize_for(self, timeout_ms: float) -> None:
        """ma_comm_synchronize_for: past the deadline the communicator is aborted and MinarrowHipError(MA_ERR_DEVICE) raised."""
        ffi.check(self.lib.ma_comm_synchronize_for(self.handle, float(timeout_ms)))

    def abort(self) -> None:
        ffi.check(self.lib.ma_comm_abort(self.handle))

    @property
    def is_broken(self) -> int:
        return int(self.lib.ma_comm_is_broken(self.handle))

    def selftest(self, timeout_ms: float, what: int = 0) -> dict:
        """ma_comm_selftest (collective: every rank calls it): the report as a dict."""
        rep = SelftestReport()
        st = self.lib.ma_comm_selftest(self.handle, int(what), float(timeout_ms), C.addressof(rep))
        out = rep.as_dict(st)
        if st != 0:
            msg = self.lib.ma_last_error_string()
            out["error"] = msg.decode() if msg else ""
        return out

    def test_stall_next_exchange(self) -> None:
        ffi.check(self.lib.ma_comm_test_stall_next_exchange(self.handle))

    def test_corrupt_next_exchange(self) -> None:
        ffi.check(self.lib.ma_comm_test_corrupt_next_exchange(self.handle))

    def close(self) -> None:
        if self.handle:
            self.lib.ma_comm_destroy(self.handle)
            self.handle = None


def arena_layout(elem_sizes, has_nulls, n_rows: int):
    """(data offsets, mask offsets or None, capacity bytes, used bytes) of the reference's arena for these columns
    (ma_arena_layout: host arithmetic only — works without a GPU)."""
    lib = ffi.load_library()
    n = len(elem_sizes)
    es = (C.c_size_t * n)(*[int(e) for e in elem_sizes])
    hn = (C.c_int32 * n)(*[1 if h else 0 for h in has_nulls])
    d_off, m_off = (C.c_size_t * n)(), (C.c_size_t * n)()
    cap, used = C.c_size_t(), C.c_size_t()
    cast = lambda a: C.cast(a, C.c_void_p)
    ffi.check(lib.ma_arena_layout(n, cast(es), cast(hn), int(n_rows), cast(d_off), cast(m_off), C.addressof(cap),
                                  C.addressof(used)))
    none = (1 << 64) - 1
    return list(d_off), [None if m == none else int(m) for m in m_off], int(cap.value), int(used.value)
