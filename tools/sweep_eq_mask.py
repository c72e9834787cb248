#!/usr/bin/env python3
"""simd_eq_mask_u{8,16,32,64} over 2^33 bytes: load depth x workgroups per CU (ctx variant bit 2048 = the round-2 shape of
4 loads per lane; blocks_per_cu overrides the grid). One JSON line per point."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

from minarrow_amd.host import Context  # noqa: E402

ctx = Context(0)
B = 1 << 33
a = ctx.alloc(B + 64)
o = ctx.alloc(B // 8 + 64)
ctx.synth_iota("i64", a, B // 8, 0)
ctx.set_async(True)
SIZE = {"u8": 1, "u16": 2, "u32": 4, "u64": 8}


import time  # noqa: E402

for tag in ("u8", "u16", "u32", "u64"):
    n = B // SIZE[tag]
    for shape, variant in (("8 loads", 0), ("4 loads", 2048)):
        for bpc in (1, 2, 3, 4, 8):
            ctx.set_variant(variant)
            ctx.set_blocks_per_cu(bpc)
            for _ in range(2):
                ctx.simd_eq_mask(tag, a, n, 0x7, 0x3, o)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                ctx.simd_eq_mask(tag, a, n, 0x7, 0x3, o)
            ctx.synchronize()
            ms = (time.perf_counter() - t0) * 100
            print(json.dumps({"type": tag, "shape": shape, "blocks_per_cu": bpc, "ms": round(ms, 4),
                              "gbps": round((B + n / 8) / ms / 1e6, 1)}), flush=True)
ctx.set_variant(0)
ctx.set_blocks_per_cu(0)
