#!/usr/bin/env python3
"""Bitmask-gated i8 / u8 sums at 4 GiB (10 % nulls): round 5's shape (8 loads per lane, two run words per lane, one workgroup
per CU) against round 4's (2 loads, three workgroups per CU; ctx variant unroll = 2), over workgroups per CU and bit offsets."""
import json, sys
sys.path.insert(0, "/root/repo")
sys.path.insert(0, ".")
from minarrow_amd.host import Context
ctx = Context(0)
nbytes = 1 << 32
a = ctx.alloc(nbytes); mask = ctx.alloc(nbytes // 8 + 64); slot = ctx.alloc(64)
ctx.synth_iota("i64", a, nbytes // 8, 1)
ctx.synth_validity(mask, nbytes, seed=5, null_every=10)
ctx.set_async(True)
r = slot.ptr
def timed(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    best = 1e9
    for _ in range(3):
        ctx.timer_start()
        for _ in range(reps): fn()
        ctx.timer_stop()
        best = min(best, ctx.timer_elapsed_ms() / reps)
    return best
for tag in ("u8", "i8"):
    for name, variant in (("deep8", 0), ("round4_2loads", 2)):
        row = {"type": tag, "shape": name}
        for bpc in (0, 1, 3):
            ctx.set_variant(variant); ctx.set_blocks_per_cu(bpc)
            for off in (13, 0):
                ms = timed(lambda: ctx.sum_into(tag, a, nbytes, out_sum=r, out_count=r + 8, mask=mask, mask_bit_offset=off))
                row[f"bpc{bpc or 'default'}_off{off}"] = round((nbytes + nbytes / 8) / ms / 1e9, 3)
        print(json.dumps(row), flush=True)
ctx.set_variant(0); ctx.set_blocks_per_cu(0)
ms = timed(lambda: ctx.sum_into("u8", a, nbytes, out_sum=r, out_count=r + 8))
print(json.dumps({"type": "u8", "shape": "dense", "tbps": round(nbytes / ms / 1e9, 3)}))
