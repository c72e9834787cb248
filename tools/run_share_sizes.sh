#!/bin/bash
# The 4-way and 2-way shares of the headline (250 M / 500 M rows per column, fused step, overlapped RCCL exchange with one rank) on one GPU:
# --scan-lanes auto (the trial's figures printed) / on / off, both launch modes. -> profiles/r05_share_sizes.txt
cd "$(dirname "$0")/.."
for rows in 250000000 500000000; do
P="--no-cpu-baseline --no-other-configs --steps 120 --warmup 10 --rows $rows --step fused"
for lanes in auto on off; do
sleep 2; python3 bench.py $P --gpus 1 --force-group --overlap on --scan-lanes $lanes 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print('group', $rows, '$lanes', round(d['ms_per_step'],4), d['config']['exchange_form'][:70], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (d['config'].get('scan_lanes_trial') or {}).items()}, [x['why'][:90] for x in d['config']['downgrades']])"
sleep 2; python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29631 bench.py $P --gpus 1 --force-dist --overlap on --scan-lanes $lanes 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print('ranks', $rows, '$lanes', round(d['ms_per_step'],4), d['config']['exchange_form'][:70], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (d['config'].get('scan_lanes_trial') or {}).items()}, [x['why'][:90] for x in d['config']['downgrades']])"
done; done
