#!/usr/bin/env python3
"""Bitmask-gated sums of every type at 4 GiB per column (10 % nulls), over workgroups per CU: the launch shape of the masked
kernels after they learnt to keep a tile requested ahead (round 4). 0 = the library's default."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402

SIZES = {"i64": 8, "u64": 8, "f64": 8, "i32": 4, "u32": 4, "f32": 4, "i16": 2, "u16": 2, "i8": 1, "u8": 1}


def timed(ctx, fn, reps=10, warm=3):
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
    nbytes = 1 << 32
    a = ctx.alloc(nbytes)
    mask = ctx.alloc(nbytes // 8 + 64)
    slot = ctx.alloc(64)
    ctx.synth_iota("i64", a, nbytes // 8, 1)
    ctx.synth_validity(mask, nbytes, seed=5, null_every=10)
    ctx.set_async(True)
    r = slot.ptr
    for tag, size in SIZES.items():
        n = nbytes // size
        row = {"type": tag, "rows": n}
        for bpc in (0, 1, 2, 3, 4):
            ctx.set_blocks_per_cu(bpc)
            if tag in ("f64", "f32"):
                fn = lambda: ctx.sum_into(tag, a, n, out_sum=r, dd_lo=r + 16, out_count=r + 8, mask=mask, mask_bit_offset=13)  # noqa: E731
            else:
                fn = lambda: ctx.sum_into(tag, a, n, out_sum=r, out_count=r + 8, mask=mask, mask_bit_offset=13)  # noqa: E731
            ms = timed(ctx, fn)
            row["bpc_%s" % (bpc or "default")] = round((nbytes + n / 8) / ms / 1e9, 3)
        ctx.set_blocks_per_cu(0)
        ms = timed(ctx, (lambda: ctx.sum_into(tag, a, n, out_sum=r, dd_lo=r + 16, out_count=r + 8)) if tag in ("f64", "f32")
                   else (lambda: ctx.sum_into(tag, a, n, out_sum=r, out_count=r + 8)))
        row["dense_default"] = round(nbytes / ms / 1e9, 3)
        print(json.dumps(row), flush=True)
    ctx.set_async(False)
    ctx.synchronize()
    ctx.close()


if __name__ == "__main__":
    main()
