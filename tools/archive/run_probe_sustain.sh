#!/bin/bash
# tools/probe_sustain.c over time and over launch shapes, with the SMI tools' view before and after. Round 4.
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/probe_sustain}
mkdir -p "$OUT"
BIN=/tmp/probe_sustain
gcc -std=gnu99 -O2 -w -Iinclude tools/probe_sustain.c -Lminarrow_amd/lib -lminarrow_hip -Wl,-rpath,$PWD/minarrow_amd/lib -o $BIN || exit 1
smi() {
    { timeout 20 rocm-smi --showpower --showtemp --showclocks --showperflevel --showmaxpower --showvoltage 2>&1
      timeout 20 amd-smi metric --power --clock --temperature --throttle 2>&1 | head -120; } > "$OUT/smi_$1.txt"
}
smi before
PROBE_TAG=timeline timeout -k 5 100 $BIN 1000000000 0 0 "10,3,4,10,4,0.5,4,1,4" > "$OUT/timeline.jsonl" 2> "$OUT/err.txt"
grep busy_s "$OUT/timeline.jsonl"
smi after_timeline
# launch shapes in the sustained state, the default shape between every two of them
: > "$OUT/shapes.jsonl"
for shape in "0 0" "0 2" "0 0" "0 3" "0 0" "8 1" "0 0" "8 2" "0 0" "4 2" "0 0" "4 4" "0 0" "32 0" "0 0" "1 0" "0 0" "16 0" "0 0"; do
    set -- $shape
    PROBE_TAG="v$1_b$2" timeout -k 5 60 $BIN 1000000000 $1 $2 "3" 2>> "$OUT/err.txt" | grep busy_s >> "$OUT/shapes.jsonl"
done
cat "$OUT/shapes.jsonl"
smi after_shapes
