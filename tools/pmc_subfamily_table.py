#!/usr/bin/env python3
"""profiles/r05_subfamily_{before,after}_*.csv (tools/pmc_subfamily.sh) -> the counter table of profiles/r05_subfamily_counters.md:
per kernel the duration, the algorithmic rate, HBM traffic against the algorithmic bytes, vector instructions per 16-byte data
load, and where the wave cycles go (SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY as fractions of SQ_WAVE_CYCLES)."""
import collections
import csv
import statistics
import sys
from pathlib import Path

P = Path(__file__).resolve().parent.parent / "profiles"
B = 1 << 32


def counters(tag, which):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(P / f"r05_subfamily_{tag}_{which}.csv")):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: statistics.median(v) for c, v in d.items()} for k, d in acc.items()}


def durations(tag):
    # the fastest of the 6 launches: the first launch of a shape in a process is up to 15 % slow (its ramp) and would own the average
    return {r["Name"]: float(r["MinNs"]) for r in csv.DictReader(open(P / f"r05_subfamily_{tag}_kernel_stats.csv"))}


def label(k):
    k = k.replace("void ma::", "").replace("unsigned char", "u8").replace("signed char", "i8").replace("unsigned short", "u16")
    k = k.replace("unsigned long", "u64").split("(")[0]
    return k


# algorithmic bytes per launch: input + what the kernel must also read (validity) or write (result bits)
ALGO = {"sum_kernel<u8, 8, false": B, "sum_kernel<u8, 2, true": B + B // 8, "sum_kernel<i8, 2, true": B + B // 8,
        "sum_kernel<u8, 8, true": B + B // 8, "sum_kernel<i8, 8, true": B + B // 8, "sum_kernel<long, 4, true": B + B // 64,
        "eq_mask_vec_kernel<u8": B + B // 8, "eq_mask_vec_kernel<u16": B + B // 16, "eq_mask_vec_kernel<u64": B + B // 64,
        "bit_scan_kernel<true>": 2 * (B // 8)}
rows = []
for tag in ("before", "after"):
    cyc, ins, fe, wr, dur = counters(tag, "cycles"), counters(tag, "insts"), counters(tag, "fetch"), counters(tag, "write"), durations(tag)
    for k in cyc:
        name = label(k)
        algo = next((v for p, v in ALGO.items() if name.startswith(p)), None)
        if algo is None:
            continue
        c, i = cyc[k], ins[k]
        data_loads = B / 1024  # 16-byte loads per lane = 1 KiB per wave instruction
        if "bit_scan" in name:
            data_loads = 2 * (B // 8) / 1024
        ns = dur[k]
        rows.append((tag, name, ns / 1e3, algo / ns / 1e3, (fe[k]["FETCH_SIZE"] * 2 + wr[k]["WRITE_SIZE"]) * 1024 / algo,
                     wr[k]["WRITE_SIZE"] * 1024 / algo, c["SQ_INSTS_VALU"] / data_loads, i["SQ_INSTS_LDS"] / data_loads,
                     c["SQ_WAVES"], c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                     c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
print("| build | kernel | us (fastest of 6) | TB/s (algorithmic) | HBM bytes / algorithmic | of which written | VALU per data load | LDS per data load | "
      "waves | issuing | issue-stalled | waiting (memory) |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print("| %s | `%s` | %.1f | %.2f | %.3f | %.3f | %.1f | %.1f | %d | %.0f %% | %.0f %% | %.0f %% |" %
          (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], 100 * r[9], 100 * r[10], 100 * r[11]))
