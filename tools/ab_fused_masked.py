#!/usr/bin/env python3
"""The fused scan with validity against the single-column masked sums, 125 M rows per column (config 5's per-GPU step: an
i64 and an f64 column of a batch sharing one validity bitmap), over workgroups per CU (round 4)."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402


def timed(ctx, fn, reps=20, warm=5):
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
    n = 125_000_000
    a, f = ctx.alloc(n * 8), ctx.alloc(n * 8)
    mask = ctx.alloc(n // 8 + 64)
    rec = ctx.alloc(256)
    ctx.synth_iota("i64", a, n, 1)
    ctx.synth_iota("f64", f, n, 1)
    ctx.synth_validity(mask, n, seed=3, null_every=10)
    ctx.set_async(True)
    r = rec.ptr
    for bpc in (0, 1, 2, 3, 4):
        ctx.set_blocks_per_cu(bpc)
        row = {"blocks_per_cu": bpc or "auto"}
        us = lambda ms: round(ms * 1e3, 1)  # noqa: E731
        row["single_i64_masked_us"] = us(timed(ctx, lambda: ctx.sum_into("i64", a, n, out_sum=r, out_count=r + 8, mask=mask)))
        row["single_f64_masked_us"] = us(timed(ctx, lambda: ctx.sum_into("f64", f, n, out_sum=r + 16, dd_lo=r + 24, out_count=r + 32, mask=mask)))
        for label, cols in (("fused_i64m", [("l", a, n, r, mask, 0)]), ("fused_f64m", [("g", f, n, r + 16, mask, 0)]),
                            ("fused_i64m_f64m", [("l", a, n, r, mask, 0), ("g", f, n, r + 16, mask, 0)]),
                            ("fused_i64_f64_dense", [("l", a, n, r), ("g", f, n, r + 16)])):
            call = ctx.prepare_sum_fused(cols)
            row[label + "_us"] = us(timed(ctx, call))
        print(json.dumps(row), flush=True)
    ctx.set_blocks_per_cu(0)
    ctx.set_async(False)
    ctx.synchronize()
    ctx.close()


if __name__ == "__main__":
    main()
