#!/usr/bin/env python3
"""Workload for a rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass: the same f64 add once with all operands 16-byte aligned
and once with lhs / rhs on other 16-byte phases (element-aligned vector loads, load16u), and a 1-byte consolidate whose
chunks start on odd bytes. HBM traffic must not grow with the misalignment (the extra cache line per wave KiB is an
L2 hit of the neighbouring wave's line)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    from minarrow_amd.host import Context

    ctx = Context(0)
    B = 1 << 30
    a, b, o = ctx.alloc(B + 256), ctx.alloc(B + 256), ctx.alloc(B + 256)
    ctx.synth_iota("f64", a, B // 8 + 8, 1)
    ctx.synth_iota("f64", b, B // 8 + 8, 3)
    n = B // 8
    for _ in range(3):
        ctx.apply("f64", a, b, 0, o, n, n)                          # aligned
    for _ in range(3):
        ctx.apply("f64", a.ptr + 8, b.ptr + 8, 0, o, n, n)          # both inputs +8 bytes
    k = 8
    per = (B // 2) // k
    chunks = [a.ptr + i * per + (i % 5) for i in range(k)]          # odd byte starts
    lens = [per - 7 - i for i in range(k)]
    for _ in range(3):
        ctx.consolidate_column(1, chunks, lens, o)
    ctx.synchronize()
    print("rows", n, "consolidated bytes", sum(lens))


if __name__ == "__main__":
    main()
