#!/bin/bash
# When should the next scan start? --variant bits 19-21 = early_mode + 1 (ma_reduce_fused.hip, FusedArgs::early_mode): the early stamp stored
# by the first workgroup to finish (mode 0), by the arrival that completes 1/4, 1/2, 3/4 of a ticket shard (1-3), or when 1 / 2 / 4 / 6 whole
# shards have arrived (4-7; 5 is the default). The 8-way share, lanes forced on (no trial), both launch modes, four processes each:
# ms per step and the span between a scan's marks (>= 0.43 ms: the overlap ran away). profiles/r05_early_mode.txt.
cd "$(dirname "$0")/.."
P="--no-cpu-baseline --no-other-configs --steps 200 --warmup 10 --rows 125000000 --step fused --overlap on --scan-lanes on"
for rep in 1 2 3 4; do
  for q in 0 3 4 5 6; do
    v=$(((q + 1) * 524288))
    sleep 1; python3 bench.py $P --gpus 1 --force-group --variant $v 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print('group mode', $q, round(d['ms_per_step'],4), round(d['kernels']['sum_fused']['span_ms']['avg'],3))"
    sleep 1; python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2963$rep bench.py $P --gpus 1 --force-dist --variant $v 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print('ranks mode', $q, round(d['ms_per_step'],4), round(d['kernels']['sum_fused']['span_ms']['avg'],3))"
  done
done
