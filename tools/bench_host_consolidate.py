#!/usr/bin/env python3
"""Host-resident SuperTable (numpy = pageable memory, as a Rust Vec64 would be) consolidated into a host arena:
the reference's bench shape (benches/consolidate.rs: 100 tables x 10 000 rows, numeric half of 20 columns) and one
large table. PCIe-inclusive wall clock; never a roofline figure."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context, arena_layout  # noqa: E402


def run(ctx, n_batches, rows, n_cols, reps):
    cells = [[(np.arange(rows, dtype=np.int64) + b * rows + c) if c % 2 == 0 else
              ((np.arange(rows, dtype=np.float64) + b * rows + c) * 0.1) for b in range(n_batches)] for c in range(n_cols)]
    _, _, capacity, _ = arena_layout([8] * n_cols, [False] * n_cols, n_batches * rows)
    arena = np.zeros(capacity + 64, dtype=np.uint8)
    base = (-arena.ctypes.data) % 64
    view = arena[base:base + capacity]
    d_off, _, _ = ctx.consolidate_table_arena([8] * n_cols, [rows] * n_batches, cells, view, capacity)
    want = np.concatenate(cells[0])
    assert np.array_equal(view[d_off[0]:d_off[0] + want.nbytes].view(np.int64), want)
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.consolidate_table_arena([8] * n_cols, [rows] * n_batches, cells, view, capacity)
        times.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for c in range(n_cols):
        np.concatenate(cells[c], out=view[d_off[c]:d_off[c] + n_batches * rows * 8].view(cells[c][0].dtype))
    numpy_s = time.perf_counter() - t0
    moved = 16 * n_cols * n_batches * rows
    return {"table": f"{n_batches} batches x {rows} rows x {n_cols} columns (host-resident)", "best_ms": min(times) * 1e3,
            "gbps_in_plus_out": moved / min(times) / 1e9, "grows_per_s": n_cols * n_batches * rows / min(times) / 1e9,
            "numpy_concatenate_one_thread_ms": numpy_s * 1e3}


if __name__ == "__main__":
    ctx = Context(0)
    out = [run(ctx, 100, 10_000, 10, 5), run(ctx, 8, 4_000_000, 4, 3)]
    print(json.dumps({"bench": "host-resident consolidate into one arena, 1 MI355X over PCIe", "results": out}))
