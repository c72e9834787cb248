#!/bin/bash
# The 8-way share (125 M rows per column, fused step, overlapped RCCL exchange with one rank) with --scan-lanes auto (two scan lanes,
# kept only if the un-timed trial measures them clearly faster; its figures are printed) against one scan stream, both launch modes,
# same box, three times. MINARROW_HIP_SCAN_LANE_CLASS=high / GPU_MAX_HW_QUEUES=8 in the environment: the A/Bs of profiles/r05_share_lanes_ab.txt.
cd "$(dirname "$0")/.."
P="--no-cpu-baseline --no-other-configs --steps 200 --warmup 10 --rows 125000000 --step fused"
for rep in 1 2 3; do
  for lanes in auto off; do
    L=$lanes
    sleep 2; python3 bench.py $P --gpus 1 --force-group --overlap on --scan-lanes $L 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print('group', '$lanes', round(d['ms_per_step'],4), d['config']['exchange_form'][:60], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (d['config'].get('scan_lanes_trial') or {}).items()}, [x['why'][:60] for x in d['config']['downgrades']])"
    sleep 2; python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2962$rep bench.py $P --gpus 1 --force-dist --overlap on --scan-lanes $L 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print('ranks', '$lanes', round(d['ms_per_step'],4), d['config']['exchange_form'][:60], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (d['config'].get('scan_lanes_trial') or {}).items()}, [x['why'][:60] for x in d['config']['downgrades']])"
  done
done
