#!/usr/bin/env python3
"""Latency of a SYNCHRONOUS call on a small column (the shape of the reference's hot-loop benches and of per-chunk
calls): enqueue + wait + result on the host. Run with MINARROW_HIP_POLL_US=0 / 60 to see what polling the completion word
in front of hipStreamSynchronize buys the reductions and the bitmap scans (null counts)."""
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402

ctx = Context(0)
out = {"MINARROW_HIP_POLL_US": os.environ.get("MINARROW_HIP_POLL_US", "(default)")}
for n in (1000, 65536, 1 << 20, 1 << 22):
    a = ctx.alloc(n * 8)
    b = ctx.alloc(n * 8)
    o = ctx.alloc(n * 8)
    ctx.synth_iota("i64", a, n, 0)
    ctx.synth_iota("i64", b, n, 1)
    for name, fn in (("sum_i64", lambda: ctx.sum("i64", a, n)), ("add_i64", lambda: ctx.apply("i64", a, b, 0, o, n, n)),
                     ("popcount_mask_bits", lambda: ctx.popcount_mask(a, 0, n))):  # the first n bits of `a` as a bitmap
        for _ in range(200):
            fn()
        reps = 3000
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        out[f"{name}_{n}_us"] = round((time.perf_counter() - t0) / reps * 1e6, 2)
    for x in (a, b, o):
        x.free()
print(json.dumps(out))
ctx.close()
