#!/usr/bin/env python3
"""Launch shape of the sum kernels for mid-size columns (2^18 .. 2^28 rows): workgroups per CU x publish form
(sc1 stores + sharded ticket vs the round-1 release/acquire fences, ctx variant bit 8). Enqueued calls timed with HIP
events on the launch stream, back to back (so each figure includes one kernel boundary). One JSON line per size."""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402


def timed(ctx, fn, reps, warm=3):
    for _ in range(warm):
        fn()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    ctx.timer_stop()
    return ctx.timer_elapsed_ms() / reps


def main():
    ctx = Context(0)
    top = 1 << 28
    a = ctx.alloc(top * 8)
    mask = ctx.alloc(top // 8 + 64)
    slot = ctx.alloc(64)
    ctx.synth_iota("f64", a, top, 0)
    ctx.synth_validity(mask, top, seed=1, null_every=10)
    ctx.set_async(True)
    bpcs = [int(x) for x in os.environ.get("MA_BPCS", "0,1,2,3,4,6,8").split(",")]
    for e in range(int(os.environ.get("MA_SWEEP_MIN", "18")), 29, 2):
        n = 1 << e
        reps = 200 if e <= 22 else (60 if e <= 26 else 20)
        row = {"rows": n}
        for name, call in (
            ("i64", lambda: ctx.sum_into("i64", a, n, out_sum=slot.ptr, out_count=slot.ptr + 8)),
            ("f64", lambda: ctx.sum_into("f64", a, n, out_sum=slot.ptr, dd_lo=slot.ptr + 16, out_count=slot.ptr + 8)),
            ("i64_masked", lambda: ctx.sum_into("i64", a, n, out_sum=slot.ptr, out_count=slot.ptr + 8, mask=mask)),
            ("f64_masked", lambda: ctx.sum_into("f64", a, n, out_sum=slot.ptr, dd_lo=slot.ptr + 16, out_count=slot.ptr + 8, mask=mask)),
        ):
            cell = {}
            for variant, tag in ((0, "sc1"), (256, "fenced")):
                ctx.set_variant(variant)
                for bpc in bpcs:
                    ctx.set_blocks_per_cu(bpc)
                    cell[f"{tag}/bpc{bpc}"] = round(timed(ctx, call, reps) * 1e3, 2)
            best = min(cell, key=cell.get)
            row[name] = {"us": cell, "best": best, "best_us": cell[best],
                         "best_gbps": round((8.125 if "masked" in name else 8) * n / cell[best] / 1e3, 1)}
        ctx.set_variant(0)
        ctx.set_blocks_per_cu(0)
        print(json.dumps(row), flush=True)
    ctx.set_async(False)
    ctx.synchronize()
    ctx.close()


if __name__ == "__main__":
    main()
