#!/bin/bash
# What state is this box's GPU in? Identity (serial / bus), throttle accumulators before and after 6 s of back-to-back 10^9-row
# sums (tools/probe_sustain.c, torch-free), sensors per sample. Appends one record to gpurun_out/box_state/. Round 4.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/box_state; mkdir -p $O
T=$(date +%H%M%S)
gcc -std=gnu99 -O2 -w -Iinclude tools/probe_sustain.c -Lminarrow_amd/lib -lminarrow_hip -Wl,-rpath,$PWD/minarrow_amd/lib -o /tmp/probe_sustain || exit 1
{ echo "== $(date -u) host $(hostname)"; timeout 20 amd-smi static --asic --bus --board 2>&1 | grep -iE "serial|bdf|product_name|market|vbios|model" | head -12
  echo "== throttle before"; timeout 20 amd-smi metric --throttle --power --temperature 2>&1 | grep -iE "socket_power|hotspot|mem:|accumul|violation" ; } > $O/$T.txt
PROBE_TAG=box_state timeout -k 5 60 /tmp/probe_sustain 1000000000 0 0 "${1:-6}" > $O/$T.jsonl 2>> $O/$T.txt
{ echo "== throttle after"; timeout 20 amd-smi metric --throttle --power --temperature 2>&1 | grep -iE "socket_power|hotspot|mem:|accumul|violation"; grep busy_s $O/$T.jsonl; } >> $O/$T.txt
cat $O/$T.txt
