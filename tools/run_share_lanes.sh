#!/bin/bash
# lanes vs one scan stream, both launch modes, same box, twice
cd "$(dirname "$0")/.."
P="--no-cpu-baseline --no-other-configs --steps 200 --warmup 10 --rows 125000000 --step fused"
for rep in 1 2; do
  for lanes in auto high off; do
    if [ $lanes = high ]; then export MINARROW_HIP_SCAN_LANE_CLASS=high; L=on; else unset MINARROW_HIP_SCAN_LANE_CLASS; L=$lanes; fi
    sleep 2; python3 bench.py $P --gpus 1 --force-group --overlap on --scan-lanes $L 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print('group', '$lanes', round(d['ms_per_step'],4), d['config']['exchange_form'][:60], d['config'].get('scan_lanes_trial'), [x['why'][:60] for x in d['config']['downgrades']])"
    sleep 2; python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2962$rep bench.py $P --gpus 1 --force-dist --overlap on --scan-lanes $L 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print('ranks', '$lanes', round(d['ms_per_step'],4), d['config']['exchange_form'][:60], d['config'].get('scan_lanes_trial'), [x['why'][:60] for x in d['config']['downgrades']])"
  done
done
