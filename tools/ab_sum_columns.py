#!/usr/bin/env python3
"""A few long columns through ma_sum_columns / ma_sum_chunks: the fused scan (default) against the general segment path
(ctx variant 16384), 8 columns of 2^26 rows and 8 of 125 M rows (config 5's batches), i64 / f64, dense / 10 % nulls, with the
plain single-column sum of the same bytes beside them (round 4)."""
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402


def timed(ctx, fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    best = 1e9
    for _ in range(3):
        ctx.timer_start()
        for _ in range(reps):
            fn()
        ctx.timer_stop()
        best = min(best, ctx.timer_elapsed_ms() / reps)
    return best


def main():
    ctx = Context(0)
    k = 8
    top = 125_000_000
    a = ctx.alloc(k * top * 8)
    mask = ctx.alloc(k * top // 8 + 64)
    outs = ctx.alloc(3 * k * 8)
    slot = ctx.alloc(64)
    ctx.synth_iota("i64", a, k * top, 1)
    ctx.synth_validity(mask, k * top, seed=3, null_every=10)
    ctx.set_async(True)
    cast = lambda arr: C.cast(arr, C.c_void_p)  # noqa: E731
    for n in (1 << 26, top):
        d = (C.c_void_p * k)(*[a.ptr + i * n * 8 for i in range(k)])
        ln = (C.c_size_t * k)(*([n] * k))
        m = (C.c_void_p * k)(*[mask.ptr + i * (n // 8) for i in range(k)])
        for fmt in "lg":
            tag = "i64" if fmt == "l" else "f64"
            for masked in (False, True):
                row = {"rows_per_column": n, "columns": k, "fmt": fmt, "masked": masked}
                bytes_ = k * n * (8.125 if masked else 8.0)
                for variant, label in ((0, "fused"), (16384, "segments")):
                    ctx.set_variant(variant)

                    def cols():
                        st = ctx.lib.ma_sum_columns(ctx.handle, ord(fmt), k, cast(d), cast(ln), cast(m) if masked else None, None,
                                                    outs.ptr, outs.ptr + 8 * k, outs.ptr + 16 * k)
                        assert st == 0, st

                    def chunks():
                        st = ctx.lib.ma_sum_chunks(ctx.handle, ord(fmt), k, cast(d), cast(ln), cast(m) if masked else None, None,
                                                   outs.ptr, outs.ptr + 8, outs.ptr + 16)
                        assert st == 0, st

                    ms = timed(ctx, cols)
                    row["sum_columns_" + label] = {"ms": round(ms, 4), "tbps": round(bytes_ / ms / 1e9, 3)}
                    ms = timed(ctx, chunks)
                    row["sum_chunks_" + label] = {"ms": round(ms, 4), "tbps": round(bytes_ / ms / 1e9, 3)}
                ctx.set_variant(0)
                ms = timed(ctx, lambda: ctx.sum_into(tag, a, k * n, out_sum=slot.ptr, out_count=slot.ptr + 8, **({"mask": mask} if masked else {}))
                           if fmt == "l" else ctx.sum_into(tag, a, k * n, out_sum=slot.ptr, dd_lo=slot.ptr + 16, out_count=slot.ptr + 8,
                                                           **({"mask": mask} if masked else {})))
                row["single_column_sum_of_the_same_bytes"] = {"ms": round(ms, 4), "tbps": round(bytes_ / ms / 1e9, 3)}
                print(json.dumps(row), flush=True)
    ctx.set_async(False)
    ctx.synchronize()
    ctx.close()


if __name__ == "__main__":
    main()
