#!/bin/bash
# The per-GPU share of the 8-way partition (125 M rows per column) on ONE GPU, every launch mode, fused and separate steps.
# A pause in front of every process: the driver clears the VRAM its predecessor released (profiles/r04_read_rate_states_root_cause.txt).
set -u
cd "$(dirname "$0")/.."
O=${1:-gpurun_out/share}; mkdir -p $O
P="--no-cpu-baseline --no-other-configs --steps 200 --warmup 10 --rows 125000000"
for s in fused separate; do
    sleep 3; python3 bench.py $P --no-torch-hosted-leg --step $s > $O/plain_$s.json 2>/dev/null
    sleep 3; python3 bench.py $P --gpus 1 --force-group --step $s > $O/group_$s.json 2>/dev/null
    sleep 3; python3 bench.py $P --gpus 1 --force-group --overlap on --step $s > $O/group_overlap_$s.json 2>/dev/null
    sleep 3; python3 bench.py $P --gpus 1 --force-group --overlap on --scan-lanes off --step $s > $O/group_overlap_one_scan_stream_$s.json 2>/dev/null
    sleep 3; python3 bench.py $P --gpus 1 --force-group --overlap on --handoff event --step $s > $O/group_overlap_event_handoff_$s.json 2>/dev/null
    sleep 3; python3 bench.py $P --gpus 1 --force-group --exchange host --step $s > $O/group_hostfold_$s.json 2>/dev/null
    sleep 3; python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py $P --gpus 1 --force-dist --overlap off --step $s > $O/ranks_$s.json 2>/dev/null
    sleep 3; python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py $P --gpus 1 --force-dist --overlap on --step $s > $O/ranks_overlap_$s.json 2>/dev/null
    sleep 3; python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29615 bench.py $P --gpus 1 --force-dist --overlap on --scan-lanes off --step $s > $O/ranks_overlap_one_scan_stream_$s.json 2>/dev/null
    sleep 3; python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29614 bench.py $P --gpus 1 --force-dist --overlap on --handoff event --step $s > $O/ranks_overlap_event_handoff_$s.json 2>/dev/null
done
python3 - $O <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f.split("/")[-1], "unreadable", e); continue
    c = d["config"]
    print(f"{f.split('/')[-1]:44s} {c.get('exchange_form', ''):58s} ms/step {d['ms_per_step']:.4f}  kernels", {k: round(v["avg_ms"], 4) for k, v in d["kernels"].items()},
          "exchange_us", round(c.get("exchange_us") or 0, 2), "fold_us", round(c.get("fold_us") or 0, 2), "host_issue_us", c.get("host_issue_us_per_step"),
          "n1", round(d.get("n1_same_process", {}).get("ms_per_step", 0), 4))
PY
