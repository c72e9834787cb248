#!/usr/bin/env python3
"""gpurun_out/profiles/<tag>_column_waves_*.csv (tools/pmc_column_waves.sh) -> a counter table in Markdown: per kernel the duration
(fastest launch), the algorithmic rate, HBM bytes fetched against the algorithmic bytes, vector / scalar / LDS instructions per
16-byte data load of a lane (= per 1 KiB a wave loads), and where the wave cycles go. argv: tag [columns] [directory]."""
import collections
import csv
import statistics
import sys
from pathlib import Path

tag = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
P = Path(sys.argv[3]) if len(sys.argv) > 3 else Path(__file__).resolve().parent.parent / "profiles"
PER = 8192


def counters(which):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(P / f"{tag}_column_waves_{which}.csv")):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: statistics.median(v) for c, v in d.items()} for k, d in acc.items()}


dur = {r["Name"]: (float(r["MinNs"]), float(r["AverageNs"]), int(r["Calls"])) for r in csv.DictReader(open(P / f"{tag}_column_waves_kernel_stats.csv"))}
cyc, ins, fe = counters("cycles"), counters("insts"), counters("fetch")


def label(k):
    return k.replace("void ma::", "").replace("(anonymous namespace)::", "").split("(")[0]


print("| kernel | us (fastest) | TB/s (algorithmic, dense bytes) | fetched / algorithmic | VALU per KiB loaded | SALU per KiB | LDS per KiB | SMEM per KiB | waves | issuing | issue-stalled | waiting |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for k in sorted(cyc):
    name = label(k)
    if not ("column_waves_kernel" in name or name.startswith("sum_kernel")) or k not in dur:
        continue
    size = 4 if "<int" in name else 8 if "<long" in name else None
    if size is None:
        continue
    algo = K * PER * size
    c, i = cyc[k], ins[k]
    kib = algo / 1024
    ns = dur[k][0]
    print("| `%s` | %.1f | %.2f | %.3f | %.1f | %.1f | %.2f | %.2f | %d | %.0f %% | %.0f %% | %.0f %% |" %
          (name, ns / 1e3, algo / ns / 1e3, fe[k]["FETCH_SIZE"] * 2 * 1024 / algo, c["SQ_INSTS_VALU"] / kib, i["SQ_INSTS_SALU"] / kib,
           i["SQ_INSTS_LDS"] / kib, i["SQ_INSTS_SMEM"] / kib, c["SQ_WAVES"], 100 * c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
           100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
