#!/usr/bin/env python3
"""ma_sum_fused against the single-column sums on the same resident columns (torch-free): one column / two columns per
launch, at the partitioned step's size (125 M rows) and at 10^9 rows, under the pace settings of ctx variant bits 5-7.
One JSON line per case: microseconds per launch (HIP events around 20 enqueued launches, best of 3) and TB/s."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402


def timed(ctx, fn, reps=20):
    for _ in range(3):
        fn()
    best = 1e9
    for _ in range(3):
        ctx.timer_start()
        for _ in range(reps):
            fn()
        ctx.timer_stop()
        best = min(best, ctx.timer_elapsed_ms() / reps)
    return best * 1e3


def main():
    ctx = Context(0)
    top = 1_000_000_000
    ci, cf = ctx.alloc(top * 8), ctx.alloc(top * 8)
    ctx.synth_iota("i64", ci, top, 0)
    ctx.synth_iota("f64", cf, top, 0)
    rec = ctx.alloc(256)
    ctx.set_async(True)
    for n in (1 << 24, 125_000_000, top):
        for variant, label in ((0, "default"), (32, "pace none"), (96, "pace 20"), (128, "pace 24")):
            ctx.set_variant(variant)
            row = {"rows": n, "pace": label}
            us = timed(ctx, lambda: ctx.sum_into("i64", ci, n, out_sum=rec.ptr, out_count=rec.ptr + 8))
            row["i64_sum"] = [round(us, 2), round(8 * n / us / 1e6, 3)]
            us = timed(ctx, lambda: ctx.sum_into("f64", cf, n, out_sum=rec.ptr + 16, dd_lo=rec.ptr + 24, out_count=rec.ptr + 32))
            row["f64_sum_dd"] = [round(us, 2), round(8 * n / us / 1e6, 3)]
            us = timed(ctx, lambda: ctx.sum_fused([("l", ci, n, rec.ptr)]))
            row["fused_i64_only"] = [round(us, 2), round(8 * n / us / 1e6, 3)]
            us = timed(ctx, lambda: ctx.sum_fused([("g", cf, n, rec.ptr + 16)]))
            row["fused_f64_only"] = [round(us, 2), round(8 * n / us / 1e6, 3)]
            us = timed(ctx, lambda: ctx.sum_fused([("l", ci, n, rec.ptr), ("g", cf, n, rec.ptr + 16)]))
            row["fused_i64_f64"] = [round(us, 2), round(16 * n / us / 1e6, 3)]

            def two():
                ctx.sum_into("i64", ci, n, out_sum=rec.ptr, out_count=rec.ptr + 8)
                ctx.sum_into("f64", cf, n, out_sum=rec.ptr + 16, dd_lo=rec.ptr + 24, out_count=rec.ptr + 32)

            us = timed(ctx, two)
            row["two_launches"] = [round(us, 2), round(16 * n / us / 1e6, 3)]
            print(json.dumps(row), flush=True)
    ctx.set_async(False)
    ctx.close()


if __name__ == "__main__":
    main()
