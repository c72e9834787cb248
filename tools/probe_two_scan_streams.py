#!/usr/bin/env python3
"""Does the NEXT step's scan start while this step's stragglers finish, if consecutive steps are launched on two alternating
streams? One stream serialises kernel k + 1 behind the last workgroup of kernel k (a 10-us spread of scan ends per 137-us scan,
plus the 1.5-us ramp: DESIGN.md §3.1); two contexts = two streams, each with its own partials and ticket. 125 M rows per column
(the 8-way share) down to 2^20 rows, single-column i64 sums over 8 distinct columns in turn, wall clock over the drained streams."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402

K = 8  # distinct columns cycled through: 8 x 2^24 rows x 8 B = 1 GiB of footprint at the smallest interesting size, far beyond any cache
ctxs = [Context(0) for _ in range(3)]
a = ctxs[0]
for rows, steps in ((1 << 20, 8000), (1 << 22, 4000), (1 << 24, 2000), (1 << 26, 600), (125_000_000, 400)):
    cols = [a.alloc(rows * 8) for _ in range(K)]
    for c in cols:
        a.synth_iota("i64", c, rows, 0)
    recs = [a.alloc(64) for _ in range(K)]
    for c in ctxs:
        c.set_async(True)
    row = {"rows_per_column": rows, "steps": steps, "distinct_columns": K}
    for lanes in (1, 2, 3, 1):
        def step(k):
            j = k % K
            ctxs[k % lanes].sum_into("i64", cols[j], rows, out_sum=recs[j].ptr, out_count=recs[j].ptr + 8)
        for k in range(3 * K):
            step(k)
        for c in ctxs:
            c.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for k in range(steps):
                step(k)
            for c in ctxs:
                c.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps * 1e3)
        key = f"{lanes}_stream{'s' if lanes > 1 else ''}_us_per_sum"
        row[key] = round(min(best * 1e3, row.get(key, 1e9)), 3)
        row[key.replace("us_per_sum", "tbps")] = round(rows * 8 / (row[key] * 1e-6) / 1e12, 3)
    print(json.dumps(row), flush=True)
    for c in ctxs:
        c.set_async(False)
    for x in (*cols, *recs):
        x.free()
