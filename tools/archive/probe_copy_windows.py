#!/usr/bin/env python3
"""Is a copy of PART of a column as fast as the same rows inside a copy of the whole? (round 4; no.)
A 999-M-row i64 column is copied into an output block whole, then window by window (10 and 20 "rounds" of 1536 x 8192 rows =
0.94 / 1.9 GiB per operand), each window repeated back to back; the same with the source window moved (no change) and the
destination window moved (the rate follows the DESTINATION window), and a pure fill of the window (the same everywhere).
Finding (profiles/r04_copy_windows.jsonl): on an output block that writes fast, most windows copy 5-10 % slower alone than
the whole block does per round — short read+write kernels lose against long ones whatever the kernel (this is the plain tile
copy) — which is why a segmented chunk list uses as few launches as keep the GPU fed (ma_consolidate.hip)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minarrow_amd.host import Context
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ab_chunked import timed
ctx = Context(0)
ctx.lib.ma_dev_output_search(1)
per, k = 8192, 122_000
n = per * k
a, b = ctx.alloc(n * 8), ctx.alloc(n * 8)
o = ctx.alloc_output(n * 8)
ctx.synth_iota("i64", a, n, 3)
ctx.set_async(True)
copy_ms = timed(ctx, lambda: ctx.consolidate_column(8, [a], [n], o))
print(json.dumps({"copy_ms": copy_ms, "copy_us_per_round": copy_ms * 1e3 / (k / 1536)}), flush=True)
RB = 1536 * per * 8  # bytes of one round of one operand
for rep in range(2):
    for R in (10, 20):
        for first in range(0, 79 - R + 1, 10):
            rows = R * 1536 * per
            reps = 200 // R
            src, dst = a.offset(first * RB), o.offset(first * RB)
            cms = timed(ctx, lambda: ctx.consolidate_column(8, [src], [rows], dst), reps=reps, warm=4)
            # the same rows of the source copied into ANOTHER window of the output, and another window of the source into this one
            other = (first + 40) % (79 - R)
            c_in = timed(ctx, lambda: ctx.consolidate_column(8, [a.offset(other * RB)], [rows], dst), reps=reps, warm=4)
            c_out = timed(ctx, lambda: ctx.consolidate_column(8, [src], [rows], o.offset(other * RB)), reps=reps, warm=4)
            wms = timed(ctx, lambda: ctx.synth_iota("i64", dst, rows, 3), reps=reps, warm=4)
            print(json.dumps({"rounds": R, "first_round": first, "copy_us_per_round": round(cms * 1e3 / R, 2),
                              "src_window_moved": round(c_in * 1e3 / R, 2), "dst_window_moved": round(c_out * 1e3 / R, 2), "other_window": other,
                              "fill_us_per_round": round(wms * 1e3 / R, 2)}), flush=True)
ctx.set_async(False)
ctx.synchronize()
ctx.close()
