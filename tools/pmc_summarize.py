#!/usr/bin/env python3
"""HBM traffic per launch from rocprofv3 --pmc passes (the guide's recipe: FETCH_SIZE and WRITE_SIZE in SEPARATE passes with
--kernel-trace only; both count KiB; on gfx950 FETCH_SIZE under-counts 16-byte-per-lane streaming loads by 2x, so it is
doubled). Writes the figures bench.py quotes as roofline.traffic:

    python tools/pmc_summarize.py <fetch_counter_collection.csv> <write_counter_collection.csv> <round> > profiles/pmc_traffic.json
"""
import csv
import json
import statistics
import sys


def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        acc.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return {k: statistics.median(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, n_f = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
rnd = sys.argv[3] if len(sys.argv) > 3 else "r03"
import datetime
import socket

out = {"collected": datetime.date.today().isoformat(), "box": f"gpurun MI355X box '{socket.gethostname()}'", "round": rnd,
       "note": f"{rnd}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1 "
               f"--no-cpu-baseline --no-other-configs` (tools/collect_profiles.sh; profiles/{rnd}_pmc_*_counter_collection.csv); "
               f"KiB, median over launches; FETCH_SIZE doubled per the gfx950 correction for 16-B/lane streaming loads; "
               f"algorithmic bytes per launch = 8e9 (16e9 for the fused i64 + f64 launch)", "kernels": {}}
for name, kib in fetch.items():
    if "sum_kernel" not in name and "sum_fused_kernel" not in name:
        continue
    b = kib * 1024 * 2
    key = "sum_fused_hbm_bytes_per_launch" if "sum_fused_kernel" in name else \
        ("sum_f64_hbm_bytes_per_launch" if "double" in name else "sum_i64_hbm_bytes_per_launch")
    if n_f[name] >= 3 and b > 1e9:  # the 10^9-row launches, not the tiny ones of set-up
        out[key] = b
    out["kernels"][name[:120]] = {"launches": n_f[name], "fetch_bytes": b, "write_bytes": write.get(name, 0.0) * 1024}
print(json.dumps(out, indent=1))
