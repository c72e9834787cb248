#!/usr/bin/env python3
"""A/B of the chunked regime's forms on ONE output block in ONE process (round 4): 122 000 x 8192-row chunks
of an 8-byte column consolidated (with / without validity) and 60 000 x 8192-row i32 chunk pairs added with nulls on both
sides — default form, tile-search form (ctx variant 128), chunk-per-workgroup form (256), one segment (1024) — each as a
fraction of the plain copy into the same block. MA_AB_SEARCH=1: take the fastest-writing of a few candidate blocks first
(ma_dev_output_search), so that the forms are compared where the store stream is not the limit."""
import ctypes as C
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402


def timed(ctx, fn, reps=5, warm=3):
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


def host_ms(ctx, fn):
    """Host time of ONE asynchronous call issued to an idle stream (what the caller's thread pays before it may go on)."""
    import time
    best = 1e9
    for _ in range(5):
        ctx.synchronize()
        t0 = time.perf_counter()
        fn()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    ctx.synchronize()
    return best


def main():
    ctx = Context(0)
    if os.environ.get("MA_AB_SEARCH"):
        ctx.lib.ma_dev_output_search(1)
    per, k = 8192, 122_000
    n = per * k
    a, b = ctx.alloc(n * 8), ctx.alloc(n * 8)
    o = ctx.alloc_output(n * 8)
    mask = ctx.alloc(n // 8 + 64)
    mask2 = ctx.alloc(n // 8 + 64)
    om = ctx.alloc(n // 8 + 64)
    ctx.synth_iota("i64", a, n, 3)
    ctx.synth_iota("i64", b, n, 7)
    ctx.synth_validity(mask, n, seed=0xC5, null_every=10)
    ctx.synth_validity(mask2, n, seed=0xC6, null_every=7)
    ctx.set_async(True)
    print(json.dumps({"output_block_write_gbps": getattr(o, "write_gbps", 0.0), "rows": n}), flush=True)
    copy_ms = timed(ctx, lambda: ctx.consolidate_column(8, [a], [n], o))
    print(json.dumps({"case": "copy (one chunk)", "ms": round(copy_ms, 4), "tbps": round(16 * n / copy_ms / 1e9, 3)}), flush=True)
    tab = lambda xs: C.cast((C.c_void_p * len(xs))(*xs), C.c_void_p)  # noqa: E731
    t_d = tab([a.ptr + i * per * 8 for i in range(k)])
    t_m = tab([mask.ptr + i * (per // 8) for i in range(k)])
    t_n = C.cast((C.c_size_t * k)(*([per] * k)), C.c_void_p)
    has = C.c_int32()

    def consolidate(masked):
        st = ctx.lib.ma_consolidate_column(ctx.handle, 8, k, t_d, t_n, t_m if masked else None, None, o.ptr, om.ptr if masked else None,
                                           C.addressof(has))
        assert st == 0, st

    for variant, label in ((0, "default"), (128, "tile-search form"), (256, "chunk-per-workgroup form"), (1024, "one segment"),
                           (256 | 1024, "chunk form, one segment")):
        ctx.set_variant(variant)
        for masked in (False, True):
            ms = timed(ctx, lambda: consolidate(masked), warm=5)
            bytes_ = (16.25 if masked else 16.0) * n
            print(json.dumps({"case": "consolidate 122000 x 8192-row i64 chunks" + (" + validity" if masked else ""), "form": label,
                              "ms": round(ms, 4), "tbps": round(bytes_ / ms / 1e9, 3), "frac_of_copy": round(copy_ms / ms * bytes_ / (16 * n), 3),
                              "host_ms_per_call": round(host_ms(ctx, lambda: consolidate(masked)), 4)}), flush=True)
    # i32 chunk pairs with nulls on both sides
    k2 = 60_000
    n2 = per * k2
    ctx.set_variant(0)  # (the loop above leaves the forced chunk form set: one chunk = one workgroup)
    copy32 = timed(ctx, lambda: ctx.consolidate_column(4, [a], [n2], o))
    l_d = tab([a.ptr + i * per * 4 for i in range(k2)])
    r_d = tab([b.ptr + i * per * 4 for i in range(k2)])
    o_d = tab([o.ptr + i * per * 4 for i in range(k2)])
    l_m = tab([mask.ptr + i * (per // 8) for i in range(k2)])
    r_m = tab([mask2.ptr + i * (per // 8) for i in range(k2)])
    o_m = tab([om.ptr + i * (per // 8) for i in range(k2)])
    n_t = C.cast((C.c_size_t * k2)(*([per] * k2)), C.c_void_p)
    hm = (C.c_int32 * k2)()

    def route(masked):
        st = ctx.lib.ma_route_super_array_broadcast(ctx.handle, ord("i"), 0, k2, l_d, n_t, l_m if masked else None, r_d, n_t,
                                                    r_m if masked else None, None, o_d, o_m if masked else None, C.cast(hm, C.c_void_p))
        assert st == 0, st

    for variant, label in ((0, "default"), (128, "tile-search form"), (128 | 64, "tile-search form, bitmaps by a second launch"),
                           (256, "chunk-per-workgroup form"), (1024, "one segment")):
        ctx.set_variant(variant)
        for masked in (False, True):
            ms = timed(ctx, lambda: route(masked), warm=5)
            bytes_ = (12.375 if masked else 12.0) * n2
            print(json.dumps({"case": "route_super_array_broadcast i32 add, 60000 x 8192-row pairs" + (", nulls on both sides" if masked else ""),
                              "form": label, "ms": round(ms, 4), "tbps": round(bytes_ / ms / 1e9, 3),
                              "frac_of_copy_rate": round((bytes_ / ms) / (8 * n2 / copy32), 3)}), flush=True)
    ctx.set_variant(0)
    ctx.set_async(False)
    ctx.synchronize()
    ctx.close()


if __name__ == "__main__":
    main()
