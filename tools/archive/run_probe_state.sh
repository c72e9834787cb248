#!/bin/bash
# One long torch-free run of back-to-back 10^9-row sums (tools/probe_sustain.c) with idle gaps, while amd-smi's full clock /
# power / temperature / throttle / usage view is sampled about once a second next to it: which sensor moves when the read
# rate steps from 7.3 to 6.9 TB/s, and how long an idle gap brings it back? Round 4.
set -u
cd "$(dirname "$0")/.."
O=${1:-gpurun_out/probe_state}; mkdir -p $O
PHASES=${2:-40,5,10,20,10,60,10}
gcc -std=gnu99 -O2 -w -Iinclude tools/probe_sustain.c -Lminarrow_amd/lib -lminarrow_hip -Wl,-rpath,$PWD/minarrow_amd/lib -o /tmp/probe_sustain || exit 1
timeout 20 amd-smi static --asic --bus --board 2>&1 | grep -iE "serial|bdf|product_name|market" > $O/box.txt
( while [ ! -e $O/stop ]; do echo "{\"epoch\": $(date +%s.%N), \"smi\": $(timeout 10 amd-smi metric --clock --power --temperature --throttle --usage --json 2>/dev/null | tr -d '\n' || echo null)}"; done > $O/smi.jsonl ) &
SAMPLER=$!
PROBE_TAG=state timeout -k 5 400 /tmp/probe_sustain 1000000000 0 0 "$PHASES" > $O/timeline.jsonl 2> $O/err.txt
touch $O/stop; wait $SAMPLER; rm -f $O/stop
grep -E "busy_s|idle_s" $O/timeline.jsonl | cut -c1-220
wc -l $O/smi.jsonl
