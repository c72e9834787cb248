#!/usr/bin/env python3
"""Does a mid-size column (<= 256 MB: the Infinity Cache's size) scan faster when the scan's loads are temporal? (round 4)
Back-to-back sums of the SAME column (the reference's own bench shape: 1000 repetitions over one vector) and sums of a
column the previous kernel has just written (fill -> sum), with non-temporal loads (the default) and temporal ones (ctx
variant bit 0), 2^22 .. 2^27 rows of i64."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402


def timed(ctx, fn, reps, warm=5):
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


def main():
    ctx = Context(0)
    top = 1 << 27
    a = ctx.alloc(top * 8)
    slot = ctx.alloc(64)
    ctx.synth_iota("i64", a, top, 0)
    ctx.set_async(True)
    for n in (1 << 22, 1 << 23, 1 << 24, 3 << 23, 1 << 25, 1 << 26, 125_000_000, 1 << 27):
        row = {"rows": n, "MiB": n * 8 >> 20}
        fill = timed(ctx, lambda: ctx.synth_iota("i64", a, n, 0), 50)
        row["fill_us"] = round(fill * 1e3, 2)
        for variant, label in ((0, "nt"), (1, "temporal")):
            ctx.set_variant(variant)
            ms = timed(ctx, lambda: ctx.sum_into("i64", a, n, out_sum=slot.ptr, out_count=slot.ptr + 8), 50)
            row["sum_" + label] = {"us": round(ms * 1e3, 2), "tbps": round(8 * n / ms / 1e9, 3)}

            def both():
                ctx.synth_iota("i64", a, n, 0)
                ctx.sum_into("i64", a, n, out_sum=slot.ptr, out_count=slot.ptr + 8)

            ms = timed(ctx, both, 50)
            row["fill_then_sum_" + label] = {"us": round(ms * 1e3, 2), "sum_part_us": round((ms - fill) * 1e3, 2)}
        ctx.set_variant(0)
        print(json.dumps(row), flush=True)
    ctx.set_async(False)
    ctx.synchronize()
    ctx.close()


if __name__ == "__main__":
    main()
