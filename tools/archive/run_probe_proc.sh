#!/bin/bash
# Runs tools/probe_proc.c many times, one process each, under the system HIP runtime, PyTorch's bundled one (preloaded,
# torch itself never imported) and the two mixed pairings, plus a few environment settings. One JSON line per process
# -> gpurun_out/probe_proc.jsonl. Round 4: what decides whether a process reads at 7.3 or at 6.9 TB/s?
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/probe_proc.jsonl}
REPS=${REPS:-5}
mkdir -p "$(dirname "$OUT")"
BIN=/tmp/probe_proc
gcc -std=gnu99 -O2 -w -Iinclude tools/probe_proc.c -Lminarrow_amd/lib -lminarrow_hip -Wl,-rpath,$PWD/minarrow_amd/lib -o $BIN || exit 1
TL=$(python3 -c "import importlib.util; print(importlib.util.find_spec('torch').submodule_search_locations[0])")/lib
SYS=/opt/rocm/lib
run() {  # tag, then VAR=value ... as the environment
    local tag=$1; shift
    env PROBE_TAG="$tag" "$@" timeout -k 5 120 $BIN >> "$OUT" 2>> "$OUT.err" || echo "{\"tag\": \"$tag\", \"error\": $?}" >> "$OUT"
}
: > "$OUT"; : > "$OUT.err"
for i in $(seq $REPS); do
    run sys
    run torch_hip+torch_hsa LD_PRELOAD="$TL/libamdhip64.so"
    run sys_hip+torch_hsa LD_PRELOAD="$TL/libhsa-runtime64.so"
    run torch_hip+sys_hsa LD_PRELOAD="$TL/libamdhip64.so $SYS/libhsa-runtime64.so.1"
    echo "round $i done: $(wc -l < "$OUT") lines"
done
for i in 1 2 3; do
    run sys+sdma0 HSA_ENABLE_SDMA=0
    run sys+hwq1 GPU_MAX_HW_QUEUES=1
    run sys+devkernarg0 HIP_FORCE_DEV_KERNARG=0
    run sys+noscratchreclaim HSA_NO_SCRATCH_RECLAIM=1
    run sys+noaslr setarch x86_64 -R
    echo "env round $i done: $(wc -l < "$OUT") lines"
done
for i in $(seq 3); do run sys_again; done
python3 - "$OUT" <<'PY'
import json, sys, collections
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip().startswith("{")]
by = collections.defaultdict(list)
for r in rows:
    by[r["tag"]].append(r.get("min"))
for t, v in by.items():
    print(f"{t:28s} " + " ".join("  err" if x is None else f"{x:5.2f}" for x in v))
PY
