#!/usr/bin/env python3
"""PCIe-inclusive rates of the same sum for host-resident columns (never the headline `value`):
  device   column in HBM (the roofline path)
  pinned   column in ma_alloc64_pinned memory (the Vec64 stand-in): kernels read it in place over PCIe
  pageable column in ordinary host memory (numpy / a Rust &[T]): staged through a temporary device buffer"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context, PinnedBuffer  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28  # 2 GiB of i64
    ctx = Context(0)
    a = np.arange(n, dtype=np.int64)
    expect = n * (n - 1) // 2
    out = {}
    dev = ctx.to_device(a)
    pin = PinnedBuffer(n * 8)
    pin.view(np.int64, n)[:] = a
    for name, buf in (("device", dev), ("pinned", pin), ("pageable", a)):
        assert ctx.sum("i64", buf, n)[0] == expect
        times = []
        for _ in range(3 if name == "pageable" else 5):
            t0 = time.perf_counter()
            s, _ = ctx.sum("i64", buf, n)
            times.append(time.perf_counter() - t0)
        assert s == expect
        best = min(times)
        out[name] = {"ms": best * 1e3, "gbps": n * 8 / best / 1e9, "grows_per_s": n / best / 1e9}
    # elementwise on pinned memory: reads and writes cross PCIe
    pout = PinnedBuffer(n * 8)
    t0 = time.perf_counter()
    ctx.apply_scalar("i64", "rhs", pin, n, 3, 0, pout)
    dt = time.perf_counter() - t0
    assert int(pout.view(np.int64, n)[12345]) == 12345 + 3
    out["pinned_add_scalar"] = {"ms": dt * 1e3, "gbps_each_way": n * 8 / dt / 1e9}
    # the same on pinned memory with the tiled pipeline off (kernel addresses host memory in place) and on (copy engines)
    pin_b = PinnedBuffer(n * 8)
    pin_b.view(np.int64, n)[:] = a[::-1]
    for label, tile in (("in_place", 0), ("tiled_32MiB", 32 << 20)):
        ctx.set_staging_tile(tile)
        for form in ("array_array", "array_scalar"):
            fn = (lambda: ctx.apply("i64", pin, pin_b, 0, pout, n, n)) if form == "array_array" else \
                (lambda: ctx.apply_scalar("i64", "rhs", pin, n, 3, 0, pout))
            fn()
            times = []
            for _ in range(3):
                t0 = time.perf_counter()
                fn()
                times.append(time.perf_counter() - t0)
            assert int(pout.view(np.int64, n)[12345]) == (n - 1 if form == "array_array" else 12345 + 3)
            moved = n * 8 * (3 if form == "array_array" else 2)
            out[f"pinned_add_{form}_{label}"] = {"ms": min(times) * 1e3, "gbps_total": moved / min(times) / 1e9,
                                                 "grows_per_s": n / min(times) / 1e9}
    # elementwise on pageable memory (a Rust &[T]): whole-operand staging vs the tiled pipeline (ma_pipeline.hip)
    b = np.arange(n, dtype=np.int64)[::-1].copy()
    res = np.zeros(n, dtype=np.int64)
    for label, tile in [("whole_operand", 0)] + [(f"tiled_{t}MiB", t << 20) for t in (8, 32)]:
        ctx.set_staging_tile(tile)
        for form in ("array_array", "array_scalar"):
            fn = (lambda: ctx.apply("i64", a, b, 0, res, n, n)) if form == "array_array" else \
                (lambda: ctx.apply_scalar("i64", "rhs", a, n, 3, 0, res))
            fn()
            times = []
            for _ in range(3):
                t0 = time.perf_counter()
                fn()
                times.append(time.perf_counter() - t0)
            assert int(res[12345]) == (n - 1 if form == "array_array" else 12345 + 3)
            moved = n * 8 * (3 if form == "array_array" else 2)
            out[f"pageable_add_{form}_{label}"] = {"ms": min(times) * 1e3, "gbps_total": moved / min(times) / 1e9,
                                                   "grows_per_s": n / min(times) / 1e9}
    ctx.set_staging_tile(32 << 20)
    print(json.dumps({"rows": n, "host_sync_wall_clock": True, **out}))


if __name__ == "__main__":
    main()
