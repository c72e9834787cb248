#!/usr/bin/env python3
"""FMA launch-shape sweep at 1B rows (f64): unroll x workgroups-per-CU."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402

n = 1_000_000_000
ctx = Context(0)
a, b, c, o = (ctx.alloc(n * 8) for _ in range(4))
for buf, s in ((a, 1), (b, 2), (c, 3)):
    ctx.synth_iota("f64", buf, n, s)
ctx.set_async(True)
best = {}
for rnd in range(3):
    for v in (2, 0, 6):
        for bpc in (1, 2, 3, 4, 8):
            ctx.set_variant(v)
            ctx.set_blocks_per_cu(bpc)
            ctx.apply_fma("f64", a, b, c, o, n, n, n)
            ctx.timer_start()
            for _ in range(5):
                ctx.apply_fma("f64", a, b, c, o, n, n, n)
            ctx.timer_stop()
            ms = ctx.timer_elapsed_ms() / 5
            best[(v, bpc)] = min(best.get((v, bpc), 1e9), ms)
for (v, bpc), ms in sorted(best.items(), key=lambda kv: kv[1]):
    print(f"fma f64 unroll={ {2: 2, 0: 4, 6: 8}[v] } bpc={bpc}  {ms:8.4f} ms  {32 * n / ms / 1e6:8.1f} GB/s  {n / ms / 1e6:6.1f} Grows/s")
