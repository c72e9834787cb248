#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv directory: runs of back-to-back dispatches of one kernel with one grid -> per run
{launches, mean kernel duration, mean gap to the next dispatch, period}, in microseconds. The first 5 of a run are skipped.
    python tools/trace_gaps.py <dir> [kernel-name filter]"""
import csv
import glob
import json
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "sum_kernel"
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    runs, cur = [], None
    for r in rows:
        key = (r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", ""))
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if cur is None or cur["key"] != key or s - cur["spans"][-1][1] > 200_000:  # a new run (or a host-side pause)
            cur = {"key": key, "spans": []}
            runs.append(cur)
        cur["spans"].append((s, e))
    for run in runs:
        name = run["key"][0]
        if flt not in name or len(run["spans"]) < 12:
            continue
        sp = run["spans"][5:]
        dur = [e - s for s, e in sp]
        gap = [sp[i + 1][0] - sp[i][1] for i in range(len(sp) - 1)]
        per = [sp[i + 1][0] - sp[i][0] for i in range(len(sp) - 1)]
        print(json.dumps({"kernel": name[:110], "grid": run["key"][1], "launches": len(sp),
                          "dur_us": round(sum(dur) / len(dur) / 1e3, 3), "dur_min_us": round(min(dur) / 1e3, 3),
                          "gap_us": round(sum(gap) / len(gap) / 1e3, 3), "gap_min_us": round(min(gap) / 1e3, 3),
                          "period_us": round(sum(per) / len(per) / 1e3, 3)}))
