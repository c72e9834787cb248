#!/usr/bin/env python3
"""consolidate of 60 000 x 8192-row chunks for every element size (+ validity), against the same-process copy."""
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402

ctx = Context(0)
k, per = 60000, 8192
a = ctx.alloc(k * per * 8 + 64)
o = ctx.alloc_output(k * per * 8 + 64)
m = ctx.alloc(k * per // 8 + 64)
om = ctx.alloc(k * per // 8 + 64)
ctx.synth_iota("i64", a, k * per, 1)
ctx.synth_validity(m, k * per, seed=3, null_every=10)
ctx.set_async(True)


def timed(fn, reps=10):
    fn(); fn(); ctx.synchronize(); fn(); ctx.timer_start()
    for _ in range(reps):
        fn()
    ctx.timer_stop()
    return ctx.timer_elapsed_ms() / reps


ms = timed(lambda: ctx.consolidate_column(8, [a], [k * per], o))
copy = 2 * k * per * 8 / ms / 1e6
print(json.dumps({"copy_gbps": round(copy, 1)}))
for elem in (1, 2, 4, 8):
    # pointer tables built once, as a host holding a SuperTable would: the timed call is the C entry point
    t_c = C.cast((C.c_void_p * k)(*[a.ptr + i * per * elem for i in range(k)]), C.c_void_p)
    t_m = C.cast((C.c_void_p * k)(*[m.ptr + i * (per // 8) for i in range(k)]), C.c_void_p)
    t_n = C.cast((C.c_size_t * k)(*([per] * k)), C.c_void_p)
    t_0 = C.cast((C.c_size_t * k)(*([0] * k)), C.c_void_p)
    has = C.c_int32()

    def call(masked):
        st = ctx.lib.ma_consolidate_column(ctx.handle, elem, k, t_c, t_n, t_m if masked else None, t_0 if masked else None, o.ptr,
                                           om.ptr if masked else None, C.addressof(has))
        assert st == 0, st

    for variant in (0, 128):
        ctx.set_variant(variant)
        ms0 = timed(lambda: call(False))
        ms1 = timed(lambda: call(True))
        ctx.set_variant(0)
        b = 2 * k * per * elem
        print(json.dumps({"elem": elem, "variant": variant, "dense_ms": round(ms0, 4), "dense_of_copy": round(b / ms0 / 1e6 / copy, 3),
                          "validity_ms": round(ms1, 4), "validity_of_copy": round((b + 2 * k * per / 8) / ms1 / 1e6 / copy, 3)}), flush=True)
