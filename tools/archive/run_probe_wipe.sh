#!/bin/bash
# tools/probe_wipe.c on this box (+ which GPU it is), then the process-level form of the same thing: probe_proc back to back
# (each starts while its predecessor's VRAM is being cleared) against probe_proc after a 2-s pause. Round 4.
set -u
cd "$(dirname "$0")/.."
O=${1:-gpurun_out/probe_wipe}; mkdir -p $O
for t in probe_proc probe_wipe; do
    gcc -std=gnu99 -O2 -w -Iinclude tools/$t.c -Lminarrow_amd/lib -lminarrow_hip -Wl,-rpath,$PWD/minarrow_amd/lib -o /tmp/$t || exit 1
done
{ timeout 20 amd-smi static --asic --bus --board --driver 2>&1 | grep -iE "serial|bdf|market|driver|version"; uname -r; cat /sys/module/amdgpu/version 2>/dev/null; } > $O/box.txt
PROBE_TAG=wipe timeout -k 5 120 /tmp/probe_wipe 0 16 0 64 16 > $O/wipe.jsonl 2> $O/err.txt
S=$O/sequence.jsonl; : > $S
p() { PROBE_TAG=$1 timeout -k 5 120 /tmp/probe_proc 1000000000 2 1 10 >> $S 2>> $O/err.txt; }
p p1; p p2_back_to_back; p p3_back_to_back; sleep 2; p p4_after_2s; p p5_back_to_back; sleep 2; p p6_after_2s; p p7_back_to_back; p p8_back_to_back
python3 - $O <<'PY'
import json, sys
d = sys.argv[1]
print(open(d + "/box.txt").read())
for l in open(d + "/wipe.jsonl"):
    r = json.loads(l)
    a = r["after"]
    dip = [x for x in a if x[1] < r["before_tbps"] - 0.15]
    print(f"ballast {r['ballast_gb']:3.0f} GB: before {r['before_tbps']:.3f}, free() {r['free_call_ms']:.1f} ms, min after {r['min_after_tbps']:.3f}, "
          f"{r['samples_0.15_below']}/{r['samples']} samples >0.15 below; dip from {dip[0][0] if dip else None} to {dip[-1][0] if dip else None} ms")
for l in open(d + "/sequence.jsonl"):
    r = json.loads(l)
    print(f"{r['tag']:20s}", r["rates_tbps"])
PY
