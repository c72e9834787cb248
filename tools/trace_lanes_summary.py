"""Reads a rocprofv3 kernel trace of bench.py's N > 1 step and says how consecutive fused scans sit on the timeline: their
duration, the interval between their starts, and by how much a scan starts before the one in front of it has ended (the overlap
two scan lanes are built for; negative = a gap). The last 150 scans of the trace — the timed loop — are summarised.
usage: trace_lanes_summary.py <label> <kernel_trace.csv> [bench line json]"""
import csv
import json
import statistics
import sys


def main():
    label, path = sys.argv[1], sys.argv[2]
    rows = list(csv.DictReader(open(path)))
    name_key = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
    scans = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows if "sum_fused_kernel" in r[name_key]]
    scans.sort()
    scans = scans[-150:]
    dur = [(e - s) / 1e3 for s, e, _ in scans]
    step = [(scans[i + 1][0] - scans[i][0]) / 1e3 for i in range(len(scans) - 1)]
    over = [(scans[i][1] - scans[i + 1][0]) / 1e3 for i in range(len(scans) - 1)]
    queues = sorted({q for _, _, q in scans})
    others = {}
    t0, t1 = scans[0][0], scans[-1][1]
    for r in rows:
        n = r[name_key]
        if "sum_fused_kernel" in n or not (t0 <= int(r["Start_Timestamp"]) <= t1):
            continue
        short = n.split("(")[0].split("<")[0][-60:]
        d = others.setdefault(short, [])
        d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {
        "scan_lanes": label,
        "scans": len(scans),
        "hardware_queues_of_the_scans": queues,
        "scan_us": {"median": round(statistics.median(dur), 1), "min": round(min(dur), 1), "max": round(max(dur), 1)},
        "start_to_start_us": {"median": round(statistics.median(step), 1), "mean": round(statistics.fmean(step), 2)},
        "next_scan_starts_before_this_one_ends_by_us": {"median": round(statistics.median(over), 1), "min": round(min(over), 1), "max": round(max(over), 1)},
        "other_kernels_in_the_window_us": {k: {"n": len(v), "median": round(statistics.median(v), 1)} for k, v in sorted(others.items())},
    }
    if len(sys.argv) > 3:
        try:
            d = json.load(open(sys.argv[3]))
            out["bench_ms_per_step_under_the_profiler"] = round(d["ms_per_step"], 4)
            out["exchange_form"] = d["config"].get("exchange_form")
        except Exception as e:  # the line is evidence, not a dependency
            out["bench_line"] = "unreadable: %s" % e
    print(json.dumps(out))


if __name__ == "__main__":
    main()
