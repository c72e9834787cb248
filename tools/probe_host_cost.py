#!/usr/bin/env python3
"""Host time per call (async context: the call returns when everything is enqueued) of the chunk-list entry points at 122 000
chunks of 8192 rows, and the steady-state time per call: sum_chunks 2.1-2.5 ns per chunk (0.25-0.30 ms) against 1.23-1.28 ms per call,
consolidate 5.9-6.5 ns against 3.1 ms, route 8.5-10 ns against 4.6-4.8 ms — every one GPU-bound back to back; the host's share
shows in ONE call from an idle stream. (Whatever entry comes first measures 1-1.7 ms on the host for its first calls: listed twice.)"""
import ctypes as C
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402

ctx = Context(0)
k, per = 122_000, 8192
n = k * per
a, b, o = ctx.alloc(n * 8 + 64), ctx.alloc(n * 8 + 64), ctx.alloc(n * 8 + 64)
m, om = ctx.alloc(n // 8 + 64), ctx.alloc(n // 8 + 64)
slot = ctx.alloc(256)
ctx.synth_iota("i64", a, n, 1)
ctx.synth_iota("i64", b, n, 3)
ctx.synth_validity(m, n, seed=1, null_every=10)
tab = lambda xs: C.cast((C.c_void_p * k)(*xs), C.c_void_p)  # noqa: E731
t_a, t_b, t_o = tab([a.ptr + i * per * 8 for i in range(k)]), tab([b.ptr + i * per * 8 for i in range(k)]), tab([o.ptr + i * per * 8 for i in range(k)])
t_m, t_om = tab([m.ptr + i * per // 8 for i in range(k)]), tab([om.ptr + i * per // 8 for i in range(k)])
t_n = C.cast((C.c_size_t * k)(*([per] * k)), C.c_void_p)
has = C.c_int32()
ctx.set_async(True)
L = ctx.lib
calls = {
    "sum_chunks masked (first)": lambda: L.ma_sum_chunks(ctx.handle, ord("l"), k, t_a, t_n, t_m, None, None, slot.ptr, slot.ptr + 8),
    "sum_chunks dense": lambda: L.ma_sum_chunks(ctx.handle, ord("l"), k, t_a, t_n, None, None, None, slot.ptr, slot.ptr + 8),
    "sum_chunks masked": lambda: L.ma_sum_chunks(ctx.handle, ord("l"), k, t_a, t_n, t_m, None, None, slot.ptr, slot.ptr + 8),
    "consolidate dense": lambda: L.ma_consolidate_column(ctx.handle, 8, k, t_a, t_n, None, None, o.ptr, None, C.addressof(has)),
    "consolidate + validity": lambda: L.ma_consolidate_column(ctx.handle, 8, k, t_a, t_n, t_m, None, o.ptr, om.ptr, C.addressof(has)),
    "route add dense": lambda: L.ma_route_super_array_broadcast(ctx.handle, ord("l"), 0, k, t_a, t_n, None, t_b, t_n, None, None, t_o, None, None),
    "route add masked": lambda: L.ma_route_super_array_broadcast(ctx.handle, ord("l"), 0, k, t_a, t_n, t_m, t_b, t_n, t_m, None, t_o, t_om, None),
}
for name, fn in calls.items():
    assert fn() == 0
    ctx.synchronize()
    hs = []
    for _ in range(5):
        ctx.synchronize()
        t0 = time.perf_counter()
        fn()
        hs.append(time.perf_counter() - t0)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    ctx.synchronize()
    steady = (time.perf_counter() - t0) / 5
    h = sorted(hs)[2]
    print(json.dumps({"call": name, "host_ms": round(h * 1e3, 3), "host_ns_per_chunk": round(h * 1e9 / k, 1), "steady_ms_per_call": round(steady * 1e3, 3)}), flush=True)
