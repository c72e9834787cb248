#!/bin/bash
# Does the 6.9 TB/s state of back-to-back short processes (tools/probe_proc.c) depend on the idle time in front of a process,
# and does a slow process recover under continuous load (tools/probe_sustain.c right behind it)? Round 4.
set -u
cd "$(dirname "$0")/.."
O=${1:-gpurun_out/probe_idle}; mkdir -p $O
for t in probe_proc probe_sustain; do
    gcc -std=gnu99 -O2 -w -Iinclude tools/$t.c -Lminarrow_amd/lib -lminarrow_hip -Wl,-rpath,$PWD/minarrow_amd/lib -o /tmp/$t || exit 1
done
S=$O/sequence.jsonl; : > $S
p() { PROBE_TAG=$1 timeout -k 5 120 /tmp/probe_proc 1000000000 2 1 ${2:-10} >> $S 2>> $O/err.txt; }
s() { PROBE_TAG=$1 timeout -k 5 120 /tmp/probe_sustain 1000000000 0 0 "$2" | grep -E "busy_s" >> $S 2>> $O/err.txt; }
p p1_first; p p2_back_to_back; p p3_back_to_back
s sustain_after_p3 "3"
p p4_after_sustain
sleep 5; p p5_after_5s_idle
sleep 5; p p6_after_5s_idle
p p7_back_to_back
sleep 15; p p8_after_15s_idle
p p9_back_to_back
p p10_bursts_of_100 100
p p11_bursts_of_100 100
p p12_bursts_of_3 3
sleep 5; p p13_bursts_of_3_after_5s_idle 3
python3 - $S <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    print(f"{r['tag']:32s}", r.get("rates_tbps") or {k: r[k] for k in ("first_tbps", "mean_tbps", "min_tbps", "max_tbps")})
PY
