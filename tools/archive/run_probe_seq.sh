#!/bin/bash
# A sequence of torch-free processes on one box: where does the 6.9 TB/s state begin (process boundary? recycled memory?),
# and which hardware counters differ between a 7.3 and a 6.9 process (TLB misses, read latency between L2 and memory)?
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
ROOT=$PWD
O=$ROOT/${1:-gpurun_out/probe_seq}; mkdir -p $O
for t in probe_proc probe_churn; do
    gcc -std=gnu99 -O2 -w -Iinclude tools/$t.c -Lminarrow_amd/lib -lminarrow_hip -Wl,-rpath,$PWD/minarrow_amd/lib -o /tmp/$t || exit 1
done
timeout 60 rocprofv3 -L > $O/counters_available.txt 2>&1
seq_out=$O/sequence.jsonl; : > $seq_out
p() { PROBE_TAG=$1 timeout -k 5 120 /tmp/probe_proc 1000000000 2 1 >> $seq_out 2>> $O/err.txt; }
p p1; p p2; p p3
PROBE_TAG=churn timeout -k 5 200 /tmp/probe_churn >> $seq_out 2>> $O/err.txt
p p4; p p5
python3 - $seq_out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    print(r["tag"], {k: v for k, v in r.items() if k in ("rates_tbps", "A_recycled", "B_six_alive", "C_after", "blocks")})
PY
cd /tmp
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_BUSY_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_GMI_sum TCC_EA0_RDREQ_IO_sum TCC_MISS_sum"; do
    i=$((i+1))
    names=""
    for c in $set; do grep -qw "$c" $O/counters_available.txt && names="$names $c"; done
    [ -z "$names" ] && { echo "pass $i: no counter of the set is available"; continue; }
    for k in 1 2; do   # two processes per pass: chances are one is in each state once the box has gone slow
        PROBE_TAG=pmc${i}_$k timeout -k 10 200 rocprofv3 --kernel-trace --pmc $names --output-format csv -d $O/pass${i}_$k -- /tmp/probe_proc 1000000000 1 1 3 > $O/pass${i}_$k.txt 2>&1 || { echo "pass $i/$k failed"; tail -3 $O/pass${i}_$k.txt; }
        cp $O/pass${i}_$k/*/*_counter_collection.csv $O/pass${i}_${k}_counters.csv 2>/dev/null
        rm -rf $O/pass${i}_$k
    done
done
cd $ROOT
python3 - $O <<'PY'
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/pass*_counters.csv")):
    agg = collections.defaultdict(list); dur = []
    for r in csv.DictReader(open(f)):
        if "sum_kernel" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if dur:
        print(f.split("/")[-1], "launches", len(dur) // max(1, len(agg)), "dur_us min %.1f median %.1f" % (min(dur), sorted(dur)[len(dur) // 2]),
              {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
