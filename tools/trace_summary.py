#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace (+ --memory-copy-trace) csv directory: consecutive dispatches of the same kernel
(the runtime's own helper kernels between them ignored, their time reported next to the group) are grouped — name, grid,
count, mean / min duration in us; memory copies are grouped by direction.
    python tools/trace_summary.py <dir> [kernel-name filter]"""
import csv
import glob
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
HELPERS = ("__amd_rocclr_", )
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    groups, helper_us = [], []
    for r in rows:
        name = r["Kernel_Name"]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if name.startswith(HELPERS):
            helper_us.append(dur)
            continue
        if flt and flt not in name:
            continue
        key = (name[:110], r.get("Grid_Size_X", r.get("Grid_Size", "")))
        if groups and groups[-1][0] == key:
            groups[-1][1].append(dur)
            groups[-1][2].extend(helper_us)
        else:
            groups.append((key, [dur], list(helper_us)))
        helper_us = []
    for (name, grid), durs, helpers in groups:
        print(f"K n={len(durs):3d} mean={sum(durs)/len(durs):9.1f}us min={min(durs):9.1f}us runtime_helper_kernels_between="
              f"{sum(helpers)/len(durs):6.1f}us/launch grid={grid} {name}")
for f in glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = {}
    for r in rows:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg.setdefault(r.get("Direction", ""), []).append(dur)
    for direction, durs in agg.items():
        print(f"C n={len(durs):4d} mean={sum(durs)/len(durs):9.1f}us max={max(durs):9.1f}us {direction}")
