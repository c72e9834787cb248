#!/usr/bin/env python3
"""<dir>/<tag>_pmc_write_lib_{counters,kernel_stats}.csv + _wall.json (tools/pmc_write_lib.sh) -> the summary table: per kernel the
duration, the bytes it moved (from the write requests themselves: 64 bytes each; a + b -> out reads two bytes per byte written, a copy
one), the rate, and the stalled share of the write requests. argv: tag directory."""
import collections, csv, json, statistics, sys
R, O = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"{O}/{R}_pmc_write_lib_counters.csv")):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {r["Name"]: (float(r["AverageNs"]), float(r["MinNs"]), int(r["Calls"])) for r in csv.DictReader(open(f"{O}/{R}_pmc_write_lib_kernel_stats.csv"))}
wall = json.load(open(f"{O}/{R}_pmc_write_lib_wall.json"))
print(f"# Write-side counters of the library's read + write kernels, {wall['rows']} f64 rows (tools/pmc_write_lib.sh): wall {json.dumps(wall['wall'])}")
print("# kernel | launches | avg us | bytes read + written per launch | TB/s | of 8 TB/s | TCC_EA0_WRREQ | 64-byte share | DRAM_CREDIT_STALL / WRREQ | WRREQ_STALL / WRREQ")
n = wall["rows"]
for k, c in acc.items():
    if k not in dur or not ("binary_vec_kernel<double" in k or "copy" in k.lower() or "concat_kernel" in k):
        continue
    m = {x: statistics.median(v) for x, v in c.items()}
    avg = dur[k][0]
    w = m.get("TCC_EA0_WRREQ_sum", 0.0)
    algo = w * 64 * (3 if "binary_vec_kernel" in k else 2)  # the runtime's copy splits a 16-GB copy into two launches
    name = k.replace("void ma::", "").split("(")[0]
    print(f"{name} | {dur[k][2]} | {avg / 1e3:.1f} | {algo:.0f} | {algo / avg / 1e3:.3f} | {algo / avg / 8e3:.3f} | {w:.0f} | "
          f"{m.get('TCC_EA0_WRREQ_64B_sum', 0.0) / w if w else 0:.3f} | {m.get('TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum', 0.0) / w if w else 0:.3f} | "
          f"{m.get('TCC_EA0_WRREQ_STALL_sum', 0.0) / w if w else 0:.3f}")
