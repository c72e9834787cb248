#!/usr/bin/env python3
"""Does the scanned / copied DATA change the rate? The write-only front does depend on it (tools/ubench_fill.hip). Here: the
10^9-row sum kernels and the a + b / copy kernels over an iota column (the reference bench's input, benches/
benchmark_parallel_simd.rs:103,115), a constant column and SplitMix64 bits (SURVEY.md 8(d)'s secondary distribution).
    python tools/bench_data_patterns.py > profiles/r03_data_patterns.json"""
import json
import os
import sys
from pathlib import Path

if os.environ.get("MA_IMPORT_TORCH"):  # A/B: PyTorch's bundled HIP runtime instead of /opt/rocm's
    import torch  # noqa: F401

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402

n = 1_000_000_000 if len(sys.argv) < 2 else int(sys.argv[1])
ctx = Context(0)
a, b, o, slot = ctx.alloc(n * 8), ctx.alloc(n * 8), ctx.alloc_output(n * 8), ctx.alloc(64)


def timed(fn, reps=10):
    fn()
    fn()
    best = 1e9
    for _ in range(2):
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        ctx.timer_stop()
        best = min(best, ctx.timer_elapsed_ms() / reps)
    return best


res = {"rows": n, "output_block_write_gbps": o.write_gbps}
ctx.set_async(True)
for name in ("iota", "constant", "splitmix"):
    for buf, seed in ((a, 1), (b, 2)):
        if name == "iota":
            ctx.synth_iota("i64", buf, n, seed)
        elif name == "constant":
            ctx.dev_memset(buf, 0x5A, n * 8)
        else:
            ctx.lib.ma_synth_splitmix_i64(ctx.handle, buf.ptr, n, 0x9E3779B97F4A7C15 * seed & ((1 << 64) - 1), 0)
    ctx.synchronize()
    e = {}
    ms = timed(lambda: ctx.sum_into("i64", a, n, out_sum=slot.ptr, out_count=slot.ptr + 8))
    e["sum_i64"] = {"ms": ms, "gbps": 8 * n / ms / 1e6}
    ms = timed(lambda: ctx.sum_into("f64", a, n, out_sum=slot.ptr, dd_lo=slot.ptr + 8, out_count=slot.ptr + 16))
    e["sum_f64_bits_as_doubles"] = {"ms": ms, "gbps": 8 * n / ms / 1e6}
    ms = timed(lambda: ctx.consolidate_column(8, [a], [n], o))
    e["copy"] = {"ms": ms, "gbps": 16 * n / ms / 1e6}
    ms = timed(lambda: ctx.apply("i64", a, b, 0, o, n, n))
    e["add_i64"] = {"ms": ms, "gbps": 24 * n / ms / 1e6}
    res[name] = e
    print(name, json.dumps(e), file=sys.stderr, flush=True)
print(json.dumps(res, indent=1))
