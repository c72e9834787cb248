#!/usr/bin/env python3
"""Throughput of EVERY kernel family of the C ABI at HBM-bound sizes, one line each — the net that catches a slow
outlier (a wrong grid, a row-at-a-time path taken by accident). Algorithmic bytes / HIP-event time on the launch
stream, inputs resident in HBM. `--bytes` is the size of one operand column (default 4 GiB).

    python tools/bench_matrix.py [--bytes 4294967296] [--reps 5] > profiles/rNN_matrix.jsonl
"""
import argparse
import json
import time
import os
import sys
from pathlib import Path

if os.environ.get("MA_IMPORT_TORCH"):  # run on PyTorch's bundled HIP runtime (what bench.py runs on) instead of /opt/rocm's
    import torch  # noqa: F401

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

SIZE = {"i8": 1, "u8": 1, "i16": 2, "u16": 2, "i32": 4, "u32": 4, "f32": 4, "i64": 8, "u64": 8, "f64": 8}
OP = {"add": 0, "sub": 1, "mul": 2, "div": 3, "rem": 4, "pow": 5, "floordiv": 6}
PEAK = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bytes", type=int, default=1 << 32)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", type=str, default="")
    args = ap.parse_args()
    from minarrow_amd.host import Context, tuning_build

    ctx = Context(0)
    print(json.dumps({"hip_runtime": [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:1]}), flush=True)
    B = args.bytes
    o = ctx.alloc_output(B + 256)  # the buffer every kernel writes: picked for its write rate (ma_dev_alloc_output)
    a, b, c = (ctx.alloc(B + 256) for _ in range(3))
    print(json.dumps({"output_block_write_gbps": round(o.write_gbps, 1)}), flush=True)
    n_bits_max = B  # one validity bit per row of the narrowest type
    mask = ctx.alloc(n_bits_max // 8 + 64)
    omask = ctx.alloc(n_bits_max // 8 + 64)
    ctx.synth_validity(mask, n_bits_max, seed=7, null_every=10)
    slot = ctx.alloc(256)

    copy_gbps = [None]

    def emit(family, tag, variant, ms, nbytes, rows):
        gbps = nbytes / ms / 1e6
        # a rate above the HBM spec means the byte accounting is wrong (aliased operands, a short-circuiting kernel)
        assert gbps <= PEAK, f"{family} {tag} {variant}: {gbps:.0f} GB/s is above the {PEAK:.0f} GB/s peak: fix the byte count"
        line = {"family": family, "type": tag, "variant": variant, "ms": round(ms, 4), "gbps": round(gbps, 1),
                "frac_of_8TBps": round(gbps / PEAK, 3), "grows_per_s": round(rows / ms / 1e6, 1)}
        if copy_gbps[0] and family not in ("sum", "mean", "sum_columns", "copy"):
            # kernels with a store stream: the same-process, same-buffers copy rate is the reference (the write rate of
            # a buffer depends on where the driver placed it: profiles/r02_probe_alloc_*.txt)
            line["frac_of_copy"] = round(gbps / copy_gbps[0], 3)
        print(json.dumps(line), flush=True)

    def timed(fn, sync=False, prime=False):
        """prime: one un-timed call immediately before the start event, so that the GPU is still busy with it while the host
        prepares the first timed call — the steady state of back-to-back calls on an async context (the host side of a
        60 000-chunk call is 0.3-0.5 ms; without this a fifth of it sits inside the events as idle GPU time)."""
        fn()
        fn()
        best = None
        for _ in range(2):  # best of two rounds: a single stalled launch (seen once: 8 ms instead of 0.9) must not be a row
            ctx.synchronize()
            if prime:
                fn()
            ctx.timer_start()
            for _ in range(args.reps):
                fn()
            ctx.timer_stop()
            ms = ctx.timer_elapsed_ms() / args.reps
            best = ms if best is None else min(best, ms)
        return best

    def want(name):
        return not args.only or any(k in name for k in args.only.split(","))

    # the reference rate of this process and these buffers: a plain 16-byte-per-lane copy a -> o
    ctx.set_async(True)
    ms = timed(lambda: ctx.consolidate_column(8, [a], [B // 8], o))
    ctx.set_async(False)
    ctx.synchronize()
    emit("copy", "8-byte", "copy kernel a -> o (reference for frac_of_copy)", ms, 2 * B, B // 8)
    copy_gbps[0] = 2 * B / ms / 1e6

    def fill(tag):
        """a = 1, 2, 3, ...; b = 3, 4, 5, ... in the widest generator that tiles the type (narrow types reinterpret)."""
        n = B // SIZE[tag]
        if tag in ("f64", "f32", "i32", "i64"):
            ctx.synth_iota(tag, a, n, 1)
            ctx.synth_iota(tag, b, n, 3)
            ctx.synth_iota(tag, c, n, 5)
        else:
            ctx.synth_iota("i64", a, B // 8, 0x0102030405060708)
            ctx.synth_iota("i64", b, B // 8, 0x0301070503010703)
        return n

    # ---- write-only kernels: the tight 1-MiB front, on the picked output block AND on a plain one (this
    # pattern writes at the same rate wherever the driver placed the block) ----
    if want("fill"):
        ctx.set_async(True)
        for name, buf in (("picked output block", o), ("plain block", c)):
            for tag in ("f64", "i32"):
                n = B // SIZE[tag]
                ms = timed(lambda: ctx.synth_iota(tag, buf, n, 1))
                emit("fill", tag, f"synth_iota into the {name}", ms, B, n)
            ms = timed(lambda: ctx.dev_memset(buf, 0xFF, B))
            emit("fill", "bytes", f"constant fill (hipMemsetAsync: constant bitmaps) into the {name}", ms, B, B)
        ctx.set_async(False)
        ctx.synchronize()

    # ---- reductions ----
    for tag in ("i64", "u64", "f64", "i32", "u32", "f32", "i16", "u16", "i8", "u8"):
        if not want("sum"):
            break
        n = fill(tag)
        ctx.set_async(True)
        ms = timed(lambda: ctx.sum_into(tag, a, n, slot.ptr, slot.ptr + 64))
        emit("sum", tag, "dense", ms, n * SIZE[tag], n)
        ms = timed(lambda: ctx.sum_into(tag, a, n, slot.ptr, slot.ptr + 64, mask=mask))
        emit("sum", tag, "masked 10% nulls", ms, n * SIZE[tag] + n / 8, n)
        ms = timed(lambda: ctx.sum_into(tag, a, n - 64, slot.ptr, slot.ptr + 64, mask=mask, mask_bit_offset=13))
        emit("sum", tag, "masked, validity at bit offset 13", ms, n * SIZE[tag] + n / 8, n)
        ctx.set_async(False)
        ctx.synchronize()

    # ---- per-column sums of many columns in two launches (ma_sum_columns): results into device slots, enqueue-only ----
    if want("sum_columns"):
        import ctypes as C
        res = ctx.alloc(3 * 8 * 65536)
        ctx.set_async(True)
        for tag, fmt in (("i64", "l"), ("f64", "g"), ("i32", "i"), ("u8", "C")):
            n = fill(tag)
            for k in (8, 1000, 60000):  # 60000: one column per RechunkStrategy::Auto chunk (8192 rows) of a chunked column
                per = (n // k) // 64 * 64 if k < 60000 else 8192
                if k == 60000 and tag == "u8":
                    continue
                ptrs = C.cast((C.c_void_p * k)(*[a.ptr + i * per * SIZE[tag] for i in range(k)]), C.c_void_p)
                lens = C.cast((C.c_size_t * k)(*([per] * k)), C.c_void_p)
                mks = C.cast((C.c_void_p * k)(*[mask.ptr + i * (per // 8) for i in range(k)]), C.c_void_p)

                def call(masked):
                    st = ctx.lib.ma_sum_columns(ctx.handle, ord(fmt), k, ptrs, lens, mks if masked else None, None, res.ptr,
                                                res.ptr + 8 * 65536 if tag != "f64" else None, res.ptr + 16 * 65536)
                    assert st == 0, st

                ms = timed(lambda: call(False), prime=k >= 1000)
                emit("sum_columns", tag, f"{k} columns of {per} rows, dense", ms, k * per * SIZE[tag], k * per)
                ms = timed(lambda: call(True), prime=k >= 1000)
                emit("sum_columns", tag, f"{k} columns of {per} rows, 10% nulls", ms, k * per * SIZE[tag] + k * per / 8, k * per)
                if k == 60000:  # the same chunk list as ONE column: a single {sum, count} (ma_sum_chunks)
                    def total(masked):
                        st = ctx.lib.ma_sum_chunks(ctx.handle, ord(fmt), k, ptrs, lens, mks if masked else None, None, res.ptr,
                                                   res.ptr + 8 * 65536 if tag != "f64" else None, res.ptr + 16 * 65536)
                        assert st == 0, st
                    ms = timed(lambda: total(False), prime=True)
                    emit("sum_chunks", tag, f"one column as {k} chunks of {per} rows, dense", ms, k * per * SIZE[tag], k * per)
                    ms = timed(lambda: total(True), prime=True)
                    emit("sum_chunks", tag, f"one column as {k} chunks of {per} rows, 10% nulls", ms, k * per * SIZE[tag] + k * per / 8, k * per)
        ctx.set_async(False)
        ctx.synchronize()
        res.free()

    # ---- elementwise ----
    for tag in SIZE:
        if not want("apply"):
            break
        n = fill(tag)
        sz = SIZE[tag]
        is_float = tag in ("f32", "f64")
        ctx.set_async(True)
        for opname in ("add", "mul") + (("div",) if tag in ("f32", "f64", "i32", "i64") else ()):
            ms = timed(lambda: ctx.apply(tag, a, b, OP[opname], o, n, n))
            emit("apply a(op)b", tag, f"{opname} dense", ms, 3 * n * sz, n)
        ms = timed(lambda: ctx.apply(tag, a, b, OP["add"], o, n, n, mask=mask, out_mask=omask))
        emit("apply a(op)b", tag, "add masked", ms, 3 * n * sz + n / 4, n)
        ms = timed(lambda: ctx.apply(tag, a.ptr + sz, b.ptr + 3 * sz, OP["add"], o, n - 64, n - 64))
        emit("apply a(op)b", tag, "add dense, operands on three different 16-byte phases", ms, 3 * n * sz, n)
        if not is_float:
            ms = timed(lambda: ctx.apply(tag, a, b, OP["div"], o, n, n, mask=mask, out_mask=omask))
            emit("apply a(op)b", tag, "div masked (validity from data)", ms, 3 * n * sz + n / 4, n)
        ms = timed(lambda: ctx.apply_scalar(tag, "rhs", a, n, 3, OP["mul"], o))
        emit("apply a(op)scalar", tag, "mul dense", ms, 2 * n * sz, n)
        ms = timed(lambda: ctx.apply_scalar(tag, "lhs", a, n, 3, OP["sub"], o, mask=mask, out_mask=omask))
        emit("apply scalar(op)a", tag, "sub masked", ms, 2 * n * sz + n / 4, n)
        if tag in ("i64", "f64", "i32", "f32"):
            ms = timed(lambda: ctx.apply_scalar(tag, "rhs", a, n, 3, OP["pow"], o))
            emit("apply a(op)scalar", tag, "pow dense", ms, 2 * n * sz, n)
            ms = timed(lambda: ctx.apply(tag, a, b, OP["floordiv"], o, n, n))
            emit("apply a(op)b", tag, "floordiv dense", ms, 3 * n * sz, n)
        if is_float:
            ms = timed(lambda: ctx.apply_fma(tag, a, b, c, o, n, n, n))
            emit("apply_fma", tag, "dense", ms, 4 * n * sz, n)
            ms = timed(lambda: ctx.apply_fma(tag, a, b, c, o, n, n, n, mask=mask, out_mask=omask))
            emit("apply_fma", tag, "masked", ms, 4 * n * sz + n / 4, n)
        ctx.set_async(False)
        ctx.synchronize()

    # ---- promotions (Int32 x Float) ----
    if want("promote"):
        n = B // 8
        ctx.synth_iota("i32", a, n, 1)
        ctx.synth_iota("f64", b, n, 3)
        ctx.set_async(True)
        ms = timed(lambda: ctx.apply_promote("i32", "f64", a, b, OP["add"], o, n, n))
        emit("apply_promote", "i32,f64", "add dense", ms, n * (4 + 8 + 8), n)
        ms = timed(lambda: ctx.apply_promote("f64", "i32", b, a, OP["mul"], o, n, n, mask=mask, out_mask=omask))
        emit("apply_promote", "f64,i32", "mul masked", ms, n * (4 + 8 + 8) + n / 4, n)
        n = B // 4
        ctx.synth_iota("f32", b, n, 3)
        ctx.synth_iota("i32", a, n, 1)
        ms = timed(lambda: ctx.apply_promote("i32", "f32", a, b, OP["add"], o, n, n))
        emit("apply_promote", "i32,f32", "add dense", ms, n * 12, n)
        ctx.set_async(False)
        ctx.synchronize()

    # ---- datetime (AND-merged masks + int kernel) ----
    if want("datetime"):
        n = B // 8
        ctx.synth_iota("i64", a, n, 1)
        ctx.synth_iota("i64", b, n, 3)
        ms = timed(lambda: ctx.apply_datetime("i64", a, 0, n, mask, b, 0, n, mask, OP["add"], o, omask))
        emit("apply_datetime", "i64", "add, both masked", ms, 3 * n * 8 + 3 * n / 8, n)

    # ---- bitmask kernels ----
    if want("bitmask"):
        nb = min(B * 8, 1 << 36)  # bits
        nb = min(nb, (B // 1) * 8)
        bits = nb
        # a, b as bitmaps of `bits` bits (already random-ish)
        ctx.synth_iota("i64", a, B // 8, 0x0102030405060708)
        ctx.synth_iota("i64", b, B // 8, 0x0301070503010703)
        for name in ("and_masks", "or_masks", "xor_masks", "eq_mask"):
            ms = timed(lambda: ctx.mask_words_op(name, a, 0, b, 0, bits, o))
            emit("bitmask", "u64 words", name, ms, 3 * bits / 8, bits)
        # in_mask: rhs holds both values here, so the result is all-true: one scan of rhs + one constant fill of out
        # (ma_in_mask, simd.rs:327-398) — 2 streams, not 3
        ms = timed(lambda: ctx.mask_words_op("in_mask", a, 0, b, 0, bits, o))
        emit("bitmask", "u64 words", "in_mask (rhs has both values: scan + fill)", ms, 2 * bits / 8, bits)
        ms = timed(lambda: ctx.mask_words_op("and_masks", a, 8, b, 24, bits - 64, o))
        emit("bitmask", "u64 words", "and_masks, byte offsets 1 and 3", ms, 3 * bits / 8, bits)
        ms = timed(lambda: ctx.mask_unary_op("not_mask", a, 0, bits, o))
        emit("bitmask", "u64 words", "not_mask", ms, 2 * bits / 8, bits)
        ms = timed(lambda: ctx.mask_unary_op("bitmask_slice", a, 13, bits - 64, o))
        emit("bitmask", "u64 words", "bitmask_slice (bit offset 13)", ms, 2 * bits / 8, bits)
        ms = timed(lambda: ctx.popcount_mask(a, 0, bits))
        emit("bitmask", "u64 words", "popcount_mask", ms, bits / 8, bits)
        ctx.dev_copy(c, a, bits // 8)  # equal contents in a DISTINCT buffer: no early exit, and two real streams
        ms = timed(lambda: ctx.mask_all("all_eq", a, 0, c, 0, bits))
        emit("bitmask", "u64 words", "all_eq (equal contents, distinct buffers)", ms, 2 * bits / 8, bits)
        ms = timed(lambda: ctx.merge_bitmasks(a, b, bits, o))
        emit("bitmask", "u64 words", "merge_bitmasks_to_new", ms, 3 * bits / 8, bits)
        for tag in ("u8", "u16", "u32", "u64"):
            n = B // SIZE[tag]
            ms = timed(lambda: ctx.simd_eq_mask(tag, a, n, 0x7, 0x3, o))
            emit("simd_eq_mask", tag, "(data & 7) == 3", ms, B + n / 8, n)

    # ---- SuperArray (op) SuperArray: all chunk pairs in one launch ----
    if want("super_array"):
        for tag, fmt in (("f64", "g"), ("i32", "i")):
            n = fill(tag)
            sz = SIZE[tag]
            ctx.set_async(True)
            for k, label in ((8, "chunks"), (min(n // 8192, 60000), "chunks (RechunkStrategy::Auto)")):
                if k == 8 and os.environ.get("MA_MATRIX_ONLY_SMALL_CHUNKS"):
                    continue  # for a kernel trace of the chunked regime alone (tools/collect_profiles.sh)
                per = (n // k) // 64 * 64 if k == 8 else 8192
                lens = [per] * k
                lhs = [a.ptr + i * per * sz for i in range(k)]
                rhs = [b.ptr + i * per * sz for i in range(k)]
                outs = [o.ptr + i * per * sz for i in range(k)]
                lms = [mask.ptr + i * (per // 8) for i in range(k)]
                oms = [omask.ptr + i * (per // 8) for i in range(k)]
                rows = per * k
                # the pointer tables are built once, as a host holding a SuperArray would: the timed call is the C entry point
                import ctypes as C
                tab = lambda xs: C.cast((C.c_void_p * k)(*xs), C.c_void_p)  # noqa: E731
                t_l, t_r, t_o, t_lm, t_om = tab(lhs), tab(rhs), tab(outs), tab(lms), tab(oms)
                t_n = C.cast((C.c_size_t * k)(*lens), C.c_void_p)
                fn = ctx.lib.ma_route_super_array_broadcast

                def call(masked):
                    st = fn(ctx.handle, ord(fmt), OP["add"], k, t_l, t_n, t_lm if masked else None, t_r, t_n,
                            t_lm if masked else None, None, t_o, t_om if masked else None, None)
                    assert st == 0, st

                for variant, vname in ((0, ""), (1024, " [one segment]"), (128, " [tile-search kernel on an uploaded table]"),
                                       (128 | 64, " [tile-search kernel, output bitmaps by a second launch]"),
                                       (256 | 16, " [chunk-per-workgroup kernel on a pinned-host table, 4 x 16 B per lane]"),
                                       (256 | 32, " [chunk-per-workgroup kernel on a pinned-host table, 8 x 16 B per lane]")):
                    if k < 2048 and variant & (256 | 1024):
                        continue  # shapes the library never picks for 8 chunks
                    if variant and not tuning_build():
                        continue  # the default matrix reports the forms the library picks; forced forms: the tuning build
                    ctx.set_variant(variant)
                    ms = timed(lambda: call(False), prime=True)
                    emit("route_super_array_broadcast", tag, f"add dense, {k} x {per}-row {label}{vname}", ms, 3 * rows * sz, rows)
                    ms = timed(lambda: call(True), prime=True)
                    emit("route_super_array_broadcast", tag, f"add, nulls on both sides, {k} x {per}-row {label}{vname}", ms, 3 * rows * sz + 3 * rows / 8, rows)
                ctx.set_variant(0)
                # SuperArray (op) Scalar (super_array.rs:87-116): the same chunks against a scalar, one launch
                import numpy as _np
                sc = _np.array([3], dtype={"f64": _np.float64, "i32": _np.int32}[tag])
                fs = ctx.lib.ma_broadcast_super_array_scalar

                def call_scalar(masked, lhs_side=0):
                    st = fs(ctx.handle, ord(fmt), OP["mul"], lhs_side, sc.ctypes.data, k, t_l, t_n, t_lm if masked else None,
                            t_o, t_om if masked else None, None)
                    assert st == 0, st

                ms = timed(lambda: call_scalar(False), prime=True)
                emit("broadcast_superarray_to_scalar", tag, f"multiply dense, {k} x {per}-row {label}", ms, 2 * rows * sz, rows)
                ms = timed(lambda: call_scalar(False, 1), prime=True)
                emit("broadcast_scalar_to_superarray", tag, f"multiply dense, {k} x {per}-row {label}", ms, 2 * rows * sz, rows)
                ms = timed(lambda: call_scalar(True), prime=True)
                emit("broadcast_superarray_to_scalar", tag, f"multiply, chunks with nulls, {k} x {per}-row {label}", ms,
                     2 * rows * sz + 2 * rows / 8, rows)
                ctx.set_variant(0)
                host = []
                for _ in range(5):  # async context: the call returns when everything is enqueued; idle stream each time
                    ctx.synchronize()
                    t0 = time.perf_counter()
                    call(True)
                    host.append((time.perf_counter() - t0) * 1e3)
                host_ms = sorted(host)[2]
                ctx.synchronize()
                for variant, vname in ((0, ""), (1024, " [one segment]"), (128, " [tile-search kernel on an uploaded table]"),
                                       (256 | 16, " [chunk-per-workgroup kernel, 4 x 16 B]"), (256 | 32, " [chunk-per-workgroup kernel, 8 x 16 B]")):
                    if variant and (k < 2048 or not tuning_build()):
                        continue
                    ctx.set_variant(variant)
                    shot = []
                    for masked in (False, True):
                        ts = []
                        for _ in range(5):  # ONE call on an idle stream, until its results are complete: host + table + kernel
                            ctx.synchronize()
                            t0 = time.perf_counter()
                            call(masked)
                            ctx.synchronize()
                            ts.append((time.perf_counter() - t0) * 1e3)
                        shot.append(round(sorted(ts)[2], 3))
                    print(json.dumps({"family": "route_super_array_broadcast", "type": tag,
                                      "variant": f"one call, idle stream to results complete, {k} chunk pairs{vname}",
                                      "wall_ms_dense": shot[0], "wall_ms_masked": shot[1],
                                      "copy_equivalent_ms": round(3 * rows * sz / copy_gbps[0] / 1e6, 3)}), flush=True)
                ctx.set_variant(0)
                print(json.dumps({"family": "route_super_array_broadcast", "type": tag, "variant": f"host time per call, {k} masked chunk pairs",
                                  "host_ms": round(host_ms, 3), "host_us_per_chunk": round(host_ms * 1e3 / k, 3)}), flush=True)
            ctx.set_async(False)
            ctx.synchronize()

    # ---- consolidate ----
    if want("consolidate"):
        for sz in (1, 2, 4, 8):
            if os.environ.get("MA_MATRIX_ONLY_SMALL_CHUNKS"):
                break
            n = (B // 2) // sz  # in + out fit the two buffers
            k = 8
            per = n // k
            chunks = [a.ptr + i * per * sz for i in range(k)]
            lens = [per - (i % 3) for i in range(k)]
            masks = [mask.ptr + i * (per // 8 // 8 * 8) for i in range(k)]
            offs = [i * 5 for i in range(k)]
            ms = timed(lambda: ctx.consolidate_column(sz, chunks, lens, o))
            emit("consolidate", f"{sz}-byte", "8 chunks, no validity", ms, 2 * sum(lens) * sz, sum(lens))
            ms = timed(lambda: ctx.consolidate_column(sz, chunks, lens, o, masks, offs, omask))
            emit("consolidate", f"{sz}-byte", "8 chunks + validity at odd bit offsets", ms, 2 * sum(lens) * sz + sum(lens) / 4, sum(lens))
        # RechunkStrategy-sized chunk lists (src/structs/chunked/super_array.rs:51-59): 60 000 chunks of 8192 rows
        import ctypes as C
        for sz in (4, 8):
            k, per = 60000, 8192
            lens = [per] * k
            tab = lambda xs: C.cast((C.c_void_p * k)(*xs), C.c_void_p)  # noqa: E731
            t_c = tab([a.ptr + i * per * sz for i in range(k)])
            t_m = tab([mask.ptr + i * (per // 8) for i in range(k)])
            t_n = C.cast((C.c_size_t * k)(*lens), C.c_void_p)
            t_o = C.cast((C.c_size_t * k)(*([0] * k)), C.c_void_p)
            fn = ctx.lib.ma_consolidate_column
            has = C.c_int32()

            def ccall(masked):
                st = fn(ctx.handle, sz, k, t_c, t_n, t_m if masked else None, t_o if masked else None, o.ptr,
                        omask.ptr if masked else None, C.addressof(has))
                assert st == 0, st

            ctx.set_async(True)
            for masked, name in ((False, "no validity"), (True, "+ validity")):
                ms = timed(lambda: ccall(masked), prime=True)
                emit("consolidate", f"{sz}-byte", f"{k} x {per}-row chunks (RechunkStrategy::Auto), {name}", ms,
                     2 * k * per * sz + (k * per / 4 if masked else 0), k * per)
                ts = []
                for _ in range(5):
                    ctx.synchronize()
                    t0 = time.perf_counter()
                    ccall(masked)
                    t1 = time.perf_counter()
                    ctx.synchronize()
                    ts.append(((time.perf_counter() - t0) * 1e3, (t1 - t0) * 1e3))
                ts.sort()
                print(json.dumps({"family": "consolidate", "type": f"{sz}-byte",
                                  "variant": f"one call, idle stream to results complete, {k} chunks, {name}",
                                  "wall_ms": round(ts[2][0], 3), "host_ms": round(ts[2][1], 3),
                                  "copy_equivalent_ms": round(2 * k * per * sz / copy_gbps[0] / 1e6, 3)}), flush=True)
            ctx.set_async(False)
            ctx.synchronize()
        bits = B * 4
        per = bits // 8
        chunks = [(a.ptr + i * (per // 64 * 8), 3 * i + 1, per - 200) for i in range(8)]
        ms = timed(lambda: ctx.consolidate_boolean_column(chunks, o))
        emit("consolidate_boolean", "bits", "8 chunks at odd bit offsets", ms, 2 * sum(c[2] for c in chunks) / 8, sum(c[2] for c in chunks))


if __name__ == "__main__":
    main()
