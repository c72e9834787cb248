#!/usr/bin/env python3
"""Grid-size resonance sweep at 1B rows: absolute workgroup counts (powers of two, CU multiples, primes) for the
multi-stream kernels (f64 add a(+)b, a(+)scalar, fma) and the sums."""
import json
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context, PinnedBuffer  # noqa: E402

n = 1_000_000_000
grids = [256, 320, 384, 448, 509, 512, 640, 761, 768, 896, 1021, 1024, 1152, 1279, 1280, 1408, 1531, 1536, 1664, 1789, 1792,
         1920, 2039, 2048, 2304, 2560, 3072, 3583, 4096]
ctx = Context(0)
a, b, c, o = (ctx.alloc(n * 8) for _ in range(4))
for buf, s in ((a, 1), (b, 2), (c, 3)):
    ctx.synth_iota("f64", buf, n, s)
slot = PinnedBuffer(64)
mask = ctx.alloc(n // 8 + 64)
ctx.synth_validity(mask, n, seed=1, null_every=10)
ctx.set_async(True)
kernels = {
    "add_aa_U8": (6, lambda: ctx.apply("f64", a, b, 0, o, n, n), 24),
    "add_aa_U4": (4, lambda: ctx.apply("f64", a, b, 0, o, n, n), 24),
    "add_as_U8": (6, lambda: ctx.apply_scalar("f64", "rhs", a, n, 2.5, 0, o), 16),
    "add_as_U4": (4, lambda: ctx.apply_scalar("f64", "rhs", a, n, 2.5, 0, o), 16),
    "fma_U4": (0, lambda: ctx.apply_fma("f64", a, b, c, o, n, n, n), 32),
    "fma_U8": (6, lambda: ctx.apply_fma("f64", a, b, c, o, n, n, n), 32),
    "sum_f64_U8": (6, lambda: ctx.sum_into("f64", a, n, out_sum=slot.ptr, out_count=slot.ptr + 8), 8),
    "sum_f64_U4": (4, lambda: ctx.sum_into("f64", a, n, out_sum=slot.ptr, out_count=slot.ptr + 8), 8),
    "sum_f64_masked_U4": (4, lambda: ctx.sum_into("f64", a, n, out_sum=slot.ptr, out_count=slot.ptr + 8, mask=mask), 8.125),
}
best = {}
for rnd in range(2):
    for name, (variant, fn, bpr) in kernels.items():
        ctx.set_variant(variant)
        for g in grids:
            ctx.set_grid(g)
            fn()
            ctx.timer_start()
            for _ in range(4):
                fn()
            ctx.timer_stop()
            ms = ctx.timer_elapsed_ms() / 4
            best[(name, g)] = min(best.get((name, g), 1e9), ms)
ctx.set_async(False)
ctx.synchronize()
rows = []
for name, (variant, fn, bpr) in kernels.items():
    line = sorted(((best[(name, g)], g) for g in grids))
    print(f"{name:18s} best: " + "  ".join(f"g={g}:{ms:.3f}ms({bpr * n / ms / 1e6:.0f}GB/s)" for ms, g in line[:5]) +
          "   worst: " + "  ".join(f"g={g}:{ms:.3f}" for ms, g in line[-3:]), flush=True)
    rows.append({"kernel": name, "bytes_per_row": bpr, "ms_by_grid": {str(g): best[(name, g)] for g in grids}})
if len(sys.argv) > 1:
    Path(sys.argv[1]).write_text(json.dumps(rows, indent=1))
