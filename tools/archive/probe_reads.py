#!/usr/bin/env python3
"""Read rate of the sum kernel on each of four 8-GB blocks a process allocates, in allocation order, two rounds
(MA_IMPORT_TORCH=1: on PyTorch's bundled HIP runtime). See profiles/r03_read_rate_by_allocation.txt."""
import os
import sys
from pathlib import Path

if os.environ.get("MA_IMPORT_TORCH"):
    import torch  # noqa: F401

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402

ctx = Context(0)
n = 1_000_000_000
bufs = [ctx.alloc(n * 8) for _ in range(4)]
slot = ctx.alloc(64)
for i, b in enumerate(bufs):
    ctx.synth_iota("i64", b, n, i)
ctx.set_async(True)


def timed(fn, reps=10):
    fn()
    fn()
    best = 1e9
    for _ in range(3):
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        ctx.timer_stop()
        best = min(best, ctx.timer_elapsed_ms() / reps)
    return best


for rnd in range(2):
    for i, b in enumerate(bufs):
        ms = timed(lambda: ctx.sum_into("i64", b, n, out_sum=slot.ptr, out_count=slot.ptr + 8))
        print(f"round {rnd} block {i} @0x{b.ptr:x}: {ms:.4f} ms = {8 * n / ms / 1e6:.0f} GB/s", flush=True)
