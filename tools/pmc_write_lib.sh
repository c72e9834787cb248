#!/bin/bash
# Write-side counters of the library's own read + write kernels (GPU box, via gpurun): one pass for the durations, one for the
# counters (no trace domain beside kernel-trace). $1 = tag. Output: gpurun_out/profiles/<tag>_pmc_write_lib_*.csv and the summary
# <tag>_pmc_write_summary.txt.
set -u
export TMPDIR=/tmp
R=${1:-r06}
O=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 tools/pmc_write_lib.py 10 > $O/${R}_pmc_write_lib_wall.json || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pw_stats -- python3 tools/pmc_write_lib.py 6 > /dev/null 2>&1 || exit 1
cp $O/pw_stats/*/*_kernel_stats.csv $O/${R}_pmc_write_lib_kernel_stats.csv; rm -rf $O/pw_stats
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/pw_pmc -- python3 tools/pmc_write_lib.py 3 > $O/${R}_pmc_write_lib.out 2>&1 || { tail -5 $O/${R}_pmc_write_lib.out; exit 1; }
cp $O/pw_pmc/*/*_counter_collection.csv $O/${R}_pmc_write_lib_counters.csv; rm -rf $O/pw_pmc
python3 - "$R" "$O" > $O/${R}_pmc_write_summary.txt <<'PY'
import collections, csv, json, statistics, sys
R, O = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"{O}/{R}_pmc_write_lib_counters.csv")):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {r["Name"]: (float(r["AverageNs"]), float(r["MinNs"]), int(r["Calls"])) for r in csv.DictReader(open(f"{O}/{R}_pmc_write_lib_kernel_stats.csv"))}
wall = json.load(open(f"{O}/{R}_pmc_write_lib_wall.json"))
print(f"# Write-side counters of the library's read + write kernels, {wall['rows']} f64 rows (tools/pmc_write_lib.sh): wall {json.dumps(wall['wall'])}")
print("# kernel | launches | avg us | algorithmic bytes | TB/s | of 8 TB/s | TCC_EA0_WRREQ | 64-byte share | DRAM_CREDIT_STALL / WRREQ | WRREQ_STALL / WRREQ")
n = wall["rows"]
for k, c in acc.items():
    if k not in dur or not ("binary_vec_kernel<double" in k or "copy" in k.lower() or "concat_kernel" in k):
        continue
    m = {x: statistics.median(v) for x, v in c.items()}
    algo = 24 * n if "binary_vec_kernel" in k else 16 * n
    avg = dur[k][0]
    w = m.get("TCC_EA0_WRREQ_sum", 0.0)
    name = k.replace("void ma::", "").split("(")[0]
    print(f"{name} | {dur[k][2]} | {avg / 1e3:.1f} | {algo} | {algo / avg / 1e3:.3f} | {algo / avg / 8e3:.3f} | {w:.0f} | "
          f"{m.get('TCC_EA0_WRREQ_64B_sum', 0.0) / w if w else 0:.3f} | {m.get('TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum', 0.0) / w if w else 0:.3f} | "
          f"{m.get('TCC_EA0_WRREQ_STALL_sum', 0.0) / w if w else 0:.3f}")
PY
cat $O/${R}_pmc_write_summary.txt
