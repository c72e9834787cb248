#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace (+ --memory-copy-trace) csv directory: one line per (kernel, grid) — dispatches,
mean / min / max duration in us, and the time of the runtime's own helper kernels (the blit that carries a table to the
device, stream-op writes) dispatched right in front of it, per launch; memory copies are grouped by direction.
    python tools/trace_summary.py <dir> [kernel-name filter]"""
import csv
import glob
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
HELPERS = ("__amd_rocclr_", )
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    groups, order, helper_us = {}, [], []
    for r in rows:
        name = r["Kernel_Name"]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if name.startswith(HELPERS):
            helper_us.append(dur)
            continue
        if flt and flt not in name:
            helper_us = []
            continue
        key = (name[:150], r.get("Grid_Size_X", r.get("Grid_Size", "")))
        if key not in groups:
            groups[key] = ([], [])
            order.append(key)
        groups[key][0].append(dur)
        groups[key][1].append(sum(helper_us))
        helper_us = []
    for key in order:
        durs, helpers = groups[key]
        print(f"K n={len(durs):4d} mean={sum(durs)/len(durs):9.1f}us min={min(durs):9.1f}us max={max(durs):9.1f}us "
              f"helper_kernels_in_front={sum(helpers)/len(durs):6.1f}us/launch grid={key[1]} {key[0]}")
for f in glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = {}
    for r in rows:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg.setdefault(r.get("Direction", ""), []).append(dur)
    for direction, durs in agg.items():
        print(f"C n={len(durs):4d} mean={sum(durs)/len(durs):9.1f}us max={max(durs):9.1f}us {direction}")
