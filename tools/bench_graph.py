#!/usr/bin/env python3
"""Launch-bound shape: K independent small sums (the reference's hot-loop benches: 1000 rows, repeated 1000 times,
benches/hotloop_benchmark_avg_simd.rs:205-208; and BASELINE configs[0]: one 10^6-row i64 sum).
Compares per-call cost of (a) synchronous calls, (b) async calls + one synchronize, (c) one hipGraph replay,
(d) ONE ma_sum_columns call for all K columns (two launches whatever K is).
Host wall clock around each variant (this is a latency measurement, not a bandwidth one)."""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import numpy as np
    from minarrow_amd.host import Context

    ctx = Context(0)
    out = []
    for n in (1000, 1_000_000):
        k = args.k
        data = ctx.alloc(k * n * 8)
        res = ctx.alloc(k * 8)
        ctx.synth_iota("i64", data, k * n, 0)
        want = np.arange(k, dtype=np.int64) * n * n + n * (n - 1) // 2

        def run_sync():
            for i in range(k):
                ctx.sum("i64", data.ptr + i * n * 8, n)

        def run_async():
            ctx.set_async(True)
            for i in range(k):
                ctx.sum_into("i64", data.ptr + i * n * 8, n, res.ptr + i * 8)
            ctx.set_async(False)
            ctx.synchronize()

        ctx.capture_begin()
        for i in range(k):
            ctx.sum_into("i64", data.ptr + i * n * 8, n, res.ptr + i * 8)
        g = ctx.capture_end()

        def run_graph():
            g.launch()

        ptrs = [data.ptr + i * n * 8 for i in range(k)]
        lens = [n] * k
        batched = {}

        def run_columns():
            batched["i64"] = ctx.sum_columns("l", ptrs, lens)[1]

        row = {"rows_per_call": n, "calls": k}
        for name, fn in (("sync_calls", run_sync), ("async_calls_one_sync", run_async), ("one_graph_replay", run_graph),
                         ("one_sum_columns_call", run_columns)):
            fn()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                fn()
            dt = (time.perf_counter() - t0) / args.reps
            row[name + "_us_per_call"] = round(dt / k * 1e6, 3)
            row[name + "_grows_per_s"] = round(k * n / dt / 1e9, 2)
        assert np.array_equal(res.download(np.int64, k), want)
        assert np.array_equal(batched["i64"], want)
        out.append(row)
        g.destroy()
        data.free()
        res.free()
    # bandwidth shape: 8 chunks x 125 M rows (config 5's per-column reduce on one device)
    k, n = 8, 125_000_000
    chunks = [ctx.alloc(n * 8) for _ in range(k)]
    for c in range(k):
        ctx.synth_iota("i64", chunks[c], n, c)
    ctx.sum_columns("l", chunks, [n] * k)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        ctx.sum_columns("l", chunks, [n] * k)
    dt = (time.perf_counter() - t0) / args.reps
    out.append({"rows_per_call": n, "calls": k, "one_sum_columns_call_ms": round(dt * 1e3, 3),
                "one_sum_columns_call_gbps": round(k * n * 8 / dt / 1e9, 1)})
    for c in chunks:
        c.free()
    # the reference's consolidate bench (benches/consolidate.rs:34-58): 100 tables x 10 000 rows, 4 and 20 columns, of
    # which the numeric half (int64 / float64 alternating) is consolidated here — column by column
    # (ma_consolidate_column: one descriptor upload + launch + synchronise each) vs the whole table into one arena
    # (ma_consolidate_table_arena: one upload, one launch, one synchronise)
    from minarrow_amd.host import arena_layout
    n_batches, rows = 100, 10_000
    for n_cols in (2, 10):
        cells = [[ctx.alloc(rows * 8) for _ in range(n_batches)] for _ in range(n_cols)]
        for c in range(n_cols):
            for b in range(n_batches):
                ctx.synth_iota("i64" if c % 2 == 0 else "f64", cells[c][b], rows, b * rows + c)
        outs = [ctx.alloc(n_batches * rows * 8) for _ in range(n_cols)]
        _, _, capacity, _ = arena_layout([8] * n_cols, [False] * n_cols, n_batches * rows)
        arena = ctx.alloc(capacity)

        def per_column():
            for c in range(n_cols):
                ctx.consolidate_column(8, cells[c], [rows] * n_batches, outs[c])

        def whole_table():
            ctx.consolidate_table_arena([8] * n_cols, [rows] * n_batches, cells, arena, capacity)

        row = {"consolidate": f"{n_batches} batches x {rows} rows x {n_cols} numeric columns", "bytes_moved": 16 * n_cols * n_batches * rows}
        for name, fn in (("per_column_calls", per_column), ("one_arena_call", whole_table)):
            fn()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                fn()
            dt = (time.perf_counter() - t0) / args.reps
            row[name + "_us"] = round(dt * 1e6, 1)
            row[name + "_gbps"] = round(row["bytes_moved"] / dt / 1e9, 1)
        s0 = ctx.sum("i64", outs[0], n_batches * rows)
        s1 = ctx.sum("i64", arena, n_batches * rows)
        assert s0 == s1, (s0, s1)
        out.append(row)
        for buf in [x for col in cells for x in col] + outs + [arena]:
            buf.free()
    print(json.dumps({"bench": "launch-bound shapes: small sums, many-batch consolidate; 1 MI355X", "results": out}))


if __name__ == "__main__":
    main()
