#!/usr/bin/env python3
"""Rate against column size, 2^10 .. 2^30 rows: where the launch floor ends and the HBM plateau begins for the sum
(read-only) and a (+) b (read+write) kernels. Enqueued calls timed with HIP events on the launch stream, inputs
resident; columns up to ~256 MiB are MALL/L2-resident across repetitions, so rates above 8 TB/s there are cache
rates, not HBM."""
import json
import os
import sys
from pathlib import Path

if os.environ.get("MA_IMPORT_TORCH"):  # A/B: the process then runs on PyTorch's bundled HIP runtime instead of /opt/rocm's
    import torch  # noqa: F401

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
    top = 1 << 30
    a, b, o = (ctx.alloc(top * 8) for _ in range(3))
    mask = ctx.alloc(top // 8 + 64)
    slot = ctx.alloc(64)
    ctx.synth_iota("f64", a, top, 0)
    ctx.synth_iota("f64", b, top, 1)
    ctx.synth_validity(mask, top, seed=1, null_every=10)
    ctx.set_async(True)
    for e in range(int(os.environ.get("MA_SWEEP_MIN", "10")), 31, 2):
        n = 1 << e
        reps = 200 if e <= 22 else (50 if e <= 26 else 10)
        row = {"rows": n, "bytes_per_operand": n * 8}
        ms = timed(ctx, lambda: ctx.sum_into("i64", a, n, out_sum=slot.ptr, out_count=slot.ptr + 8), reps)
        row["sum_i64"] = {"us": ms * 1e3, "gbps": 8 * n / ms / 1e6}
        ms = timed(ctx, lambda: ctx.sum_into("f64", a, n, out_sum=slot.ptr, dd_lo=slot.ptr + 16, out_count=slot.ptr + 8), reps)
        row["sum_f64"] = {"us": ms * 1e3, "gbps": 8 * n / ms / 1e6}
        ms = timed(ctx, lambda: ctx.sum_into("i64", a, n, out_sum=slot.ptr, out_count=slot.ptr + 8, mask=mask), reps)
        row["sum_i64_masked"] = {"us": ms * 1e3, "gbps": 8.125 * n / ms / 1e6}
        ms = timed(ctx, lambda: ctx.apply("f64", a, b, 0, o, n, n), reps)
        row["add_f64"] = {"us": ms * 1e3, "gbps": 24 * n / ms / 1e6}
        ms = timed(ctx, lambda: ctx.apply_scalar("f64", "rhs", a, n, 2.5, 2, o), reps)
        row["mul_f64_scalar"] = {"us": ms * 1e3, "gbps": 16 * n / ms / 1e6}
        print(json.dumps(row), flush=True)
    ctx.set_async(False)
    ctx.synchronize()
    ctx.close()
    print(json.dumps({"hip_runtime": [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:1]}), flush=True)


if __name__ == "__main__":
    main()
