#!/usr/bin/env python3
"""Back-to-back dense i64 / f64 sums at the mid sizes (2^10 .. 2^27 rows, 125 M rows), 60 launches each, for a
rocprofv3 --kernel-trace run: tools/trace_gaps.py then splits every size's period into kernel duration and the gap to the
next dependent dispatch. Torch-free (the system HIP runtime)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402

SIZES = [1 << 10, 1 << 16, 1 << 20, 1 << 22, 1 << 24, 1 << 25, 1 << 26, 125_000_000, 1 << 27]


def main():
    ctx = Context(0)
    top = max(SIZES)
    a = ctx.alloc(top * 8)
    slot = ctx.alloc(64)
    ctx.synth_iota("f64", a, top, 0)
    ctx.set_async(True)
    for tag in ("i64", "f64"):
        for n in SIZES:
            for _ in range(60):
                if tag == "i64":
                    ctx.sum_into("i64", a, n, out_sum=slot.ptr, out_count=slot.ptr + 8)
                else:
                    ctx.sum_into("f64", a, n, out_sum=slot.ptr, dd_lo=slot.ptr + 16, out_count=slot.ptr + 8)
            ctx.synchronize()
    ctx.set_async(False)
    ctx.close()


if __name__ == "__main__":
    main()
