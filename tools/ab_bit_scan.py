#!/usr/bin/env python3
"""The bitmap scans (popcount / null count, all_true, all_eq) over bitmap sizes, call to result on the host (round 4: the
scan's cross-workgroup step became the sums' partial + ticket hand-off instead of four atomics per workgroup)."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402


def main():
    ctx = Context(0)
    top = 1 << 35  # bits: 4 GiB
    a, b = ctx.alloc(top // 8 + 64), ctx.alloc(top // 8 + 64)
    ctx.synth_validity(a, top, seed=5, null_every=10)
    ctx.synth_validity(b, top, seed=5, null_every=10)
    for bpc in (0, 1, 2, 3):
        ctx.set_blocks_per_cu(bpc)
        for e in (10, 20, 24, 28, 30, 32, 35):
            n = 1 << e
            row = {"blocks_per_cu": bpc or "default", "log2_bits": e, "bytes": n // 8}
            for name, fn in (("popcount", lambda: ctx.popcount_mask(a, 0, n)), ("all_eq", lambda: ctx.mask_all("all_eq", a, 0, b, 0, n))):
                for _ in range(3):
                    fn()
                reps = 200 if e <= 24 else 20
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                us = (time.perf_counter() - t0) / reps * 1e6
                streams = 1 if name == "popcount" else 2
                row[name] = {"us_call_to_result": round(us, 2), "tbps": round(streams * n / 8 / us / 1e6, 3)}
            print(json.dumps(row), flush=True)
    ctx.set_blocks_per_cu(0)
    ctx.close()


if __name__ == "__main__":
    main()
