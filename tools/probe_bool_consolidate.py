#!/usr/bin/env python3
"""Boolean consolidate (bit-granular join) of many small chunks: 2^34 bits as 8 chunks, as 2^21 x 8192-bit chunks (a Boolean
column rechunked at RechunkStrategy::Auto's 8192 rows) aligned and at odd bit offsets, against the same-process copy."""
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402

ctx = Context(0)
BITS = 1 << 33
a = ctx.alloc(BITS // 8 + 4096)
o = ctx.alloc_output(BITS // 8 + 64)
ctx.synth_iota("i64", a, BITS // 64, 0x0123456789ABCDEF)
ctx.set_async(True)


def timed(fn, reps=5):
    fn(); fn(); ctx.synchronize(); fn(); ctx.timer_start()
    for _ in range(reps):
        fn()
    ctx.timer_stop()
    return ctx.timer_elapsed_ms() / reps


ms = timed(lambda: ctx.consolidate_column(8, [a], [BITS // 64], o))
copy = 2 * (BITS // 8) / ms / 1e6
print(json.dumps({"copy_gbps": round(copy, 1)}))
for k, per, off, label in ((8, BITS // 8, 3, "8 chunks at bit offset 3"), (BITS // 8192 // 8, 8192, 0, "131072 x 8192-bit chunks, aligned"),
                           (BITS // 8192 // 8, 8192 - 3, 5, "131072 x 8189-bit chunks at bit offset 5")):
    stride = (8192 // 8) if k > 8 else per // 8
    t_c = C.cast((C.c_void_p * k)(*[a.ptr + i * stride for i in range(k)]), C.c_void_p)
    t_o = C.cast((C.c_size_t * k)(*([off] * k)), C.c_void_p)
    t_n = C.cast((C.c_size_t * k)(*([per] * k)), C.c_void_p)
    has = C.c_int32()

    def call():
        st = ctx.lib.ma_consolidate_boolean_column(ctx.handle, k, t_c, t_o, t_n, None, None, o.ptr, None, C.addressof(has))
        assert st == 0, st

    ms = timed(call)
    import time
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        call()
    host_ms = (time.perf_counter() - t0) / 5 * 1e3  # async context: the call returns when everything is enqueued
    ctx.synchronize()
    b = 2 * k * per / 8
    print(json.dumps({"shape": label, "ms": round(ms, 4), "gbps": round(b / ms / 1e6, 1), "of_copy": round(b / ms / 1e6 / copy, 3),
                      "host_ms_per_call": round(host_ms, 4)}), flush=True)
