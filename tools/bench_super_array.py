#!/usr/bin/env python3
"""SuperArray (op) SuperArray at the reference's default chunking (8192 rows, RechunkStrategy::Auto): 10^9 f64 rows =
122 071 chunk pairs in ONE launch, vs the same column as a single array."""
import json
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
chunk = 8192
ctx = Context(0)
a, b, o = (ctx.alloc(n * 8) for _ in range(3))
ctx.synth_iota("f64", a, n, 1)
ctx.synth_iota("f64", b, n, 2)
k = (n + chunk - 1) // chunk
lens = [min(chunk, n - i * chunk) for i in range(k)]
lhs = [a.ptr + i * chunk * 8 for i in range(k)]
rhs = [b.ptr + i * chunk * 8 for i in range(k)]
outs = [o.ptr + i * chunk * 8 for i in range(k)]
t0 = time.perf_counter()
ctx.route_super_array_broadcast("g", 0, lhs, rhs, lens, lens, outs)
first = time.perf_counter() - t0
times = []
for _ in range(3):
    t0 = time.perf_counter()
    ctx.route_super_array_broadcast("g", 0, lhs, rhs, lens, lens, outs)
    times.append(time.perf_counter() - t0)
s, _ = ctx.sum("f64", o, n)
ctx.set_async(True)
ctx.apply("f64", a, b, 0, o, n, n)
ctx.timer_start()
for _ in range(5):
    ctx.apply("f64", a, b, 0, o, n, n)
ctx.timer_stop()
single_ms = ctx.timer_elapsed_ms() / 5
print(json.dumps({"rows": n, "chunks": k, "chunk_rows": chunk, "host_wall_ms_first": first * 1e3, "host_wall_ms_best": min(times) * 1e3,
                  "grows_per_s_incl_host": n / min(times) / 1e9, "single_array_kernel_ms": single_ms,
                  "sum_check": s == float(n) * (n + 1) / 2 + float(n) * (n + 3) / 2}))
