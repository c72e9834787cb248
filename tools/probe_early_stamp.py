#!/usr/bin/env python3
"""Consecutive fused scans on TWO alternating contexts, the next one gated on the previous one's EARLY stamp (stored by every
workgroup as soon as its rows are scanned: the first to finish gets there first) instead of running fully concurrently: the next
scan's ramp then runs under this scan's stragglers and hand-off only. Forms, per column size, i64 + f64 fused step over K distinct
column pairs: one context; two contexts free-running (tools/probe_two_scan_streams.py's form); two contexts gated on the early
stamp; two contexts gated on the FINAL stamp (= one stream's order, on two streams: the control)."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402

import os  # noqa: E402

K = 4
ctxs = [Context(0)]
if len(sys.argv) > 1:  # "high" / "low": the second context's stream in another priority class (its own hardware-queue pool)
    os.environ["MINARROW_HIP_STREAM_PRIORITY"] = sys.argv[1]
ctxs.append(Context(0))
os.environ.pop("MINARROW_HIP_STREAM_PRIORITY", None)
a = ctxs[0]
stamps = [c.stamp_alloc() for c in ctxs]
seq = [0, 0]
for rows, steps in ((1 << 24, 2000), (1 << 26, 600), (125_000_000, 400), (250_000_000, 200), (1_000_000_000, 60))[:int(os.environ.get("PROBE_SIZES", "5"))]:
    pairs = [(a.alloc(rows * 8), a.alloc(rows * 8)) for _ in range(K if rows < 500_000_000 else 2)]
    for ci, cf in pairs:
        a.synth_iota("i64", ci, rows, 0)
        a.synth_iota("f64", cf, rows, 0)
    recs = [a.alloc(256) for _ in range(2 * len(pairs))]
    for c in ctxs:
        c.set_async(True)
    # stamped launches; the "early" form additionally stores the value to word 1 of the stamp's line as soon as a workgroup is done
    calls = {early: [[c.prepare_sum_fused([("l", ci, rows, recs[2 * j + k].ptr), ("g", cf, rows, recs[2 * j + k].ptr + 16)], stamp=stamps[k],
                                          early=(stamps[k] + 8) if early else 0)
                      for j, (ci, cf) in enumerate(pairs)] for k, c in enumerate(ctxs)] for early in (False, True)}
    row = {"rows_per_column": rows, "steps": steps}

    def run(form):
        def step(k):
            lane = 0 if form == "one" else k & 1
            if form in ("early", "final") and k > 0:
                other = lane ^ 1
                ctxs[lane].wait_value(stamps[other] + (8 if form == "early" else 0), seq[other])
            seq[lane] += 1
            calls[form == "early"][lane][k % len(pairs)](seq[lane])
        for k in range(8):
            step(k)
        for c in ctxs:
            c.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for k in range(steps):
                step(k)
            for c in ctxs:
                c.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps * 1e3)
        return best

    for form in ("one", "free", "early", "final", "one"):
        ms = run(form)
        row[form + "_ms"] = round(min(ms, row.get(form + "_ms", 1e9)), 5)
    for form in ("free", "early", "final"):
        row[form + "_vs_one"] = round(row[form + "_ms"] / row["one_ms"], 4)
    row["tbps_one"] = round(rows * 16 / row["one_ms"] / 1e9, 3)
    row["tbps_early"] = round(rows * 16 / row["early_ms"] / 1e9, 3)
    print(json.dumps(row), flush=True)
    for c in ctxs:
        c.set_async(False)
    for x in (*[c for p in pairs for c in p], *recs):
        x.free()
