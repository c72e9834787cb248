#!/usr/bin/env python3
"""Launch shapes of the narrow-integer sums (i8 / i16): unroll (variant bits 1-3) x workgroups per CU, dense and masked."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402

ctx = Context(0)
B = 1 << 32
buf, mask, slot = ctx.alloc(B + 256), ctx.alloc(B // 8 + 64), ctx.alloc(256)
ctx.synth_iota("i64", buf, B // 8, 0x0102030405060708)
ctx.synth_validity(mask, B, seed=7, null_every=10)
ctx.set_async(True)
for tag, size in (("i8", 1), ("i16", 2)):
    n = B // size
    for masked in (False, True):
        for variant in (0, 2, 4, 6):  # unroll auto / 2 / 4 / 8
            for bpc in (0, 1, 2, 3, 4, 6, 8):
                ctx.set_variant(variant)
                ctx.set_blocks_per_cu(bpc)
                fn = lambda: ctx.sum_into(tag, buf, n, slot.ptr, slot.ptr + 64, mask=mask if masked else None)
                fn(); fn()
                best = 1e9
                for _ in range(2):
                    ctx.synchronize(); ctx.timer_start()
                    for _ in range(5): fn()
                    ctx.timer_stop(); best = min(best, ctx.timer_elapsed_ms() / 5)
                print(json.dumps({"type": tag, "masked": masked, "unroll": {0: "auto", 2: 2, 4: 4, 6: 8}[variant], "bpc": bpc,
                                  "ms": round(best, 4), "frac": round((B + (n / 8 if masked else 0)) / best / 1e6 / 8000, 3)}), flush=True)
