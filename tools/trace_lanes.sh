#!/bin/bash
# What two scan lanes do to the timeline, from the profiler's own timestamps: rocprofv3 --kernel-trace (the program itself after `--`,
# no counters) of the 8-way share in the one-process group mode (125 M rows per column, fused step, overlapped RCCL exchange with one
# rank), once with the lanes and once on one scan stream; tools/trace_lanes_summary.py reads the kernel traces.
# -> profiles/r05_lanes_kernel_trace.txt
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
O=gpurun_out/trace_lanes
rm -rf $O; mkdir -p $O
P="--no-cpu-baseline --no-other-configs --no-torch-hosted-leg --steps 200 --warmup 10 --rows 125000000 --step fused --gpus 1 --force-group --overlap on"
for lanes in on off; do
  sleep 3
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/$lanes -- python3 bench.py $P --scan-lanes $lanes > $O/bench_$lanes.json 2> $O/bench_$lanes.err || exit 1
  python3 tools/trace_lanes_summary.py $lanes $O/$lanes/*/*_kernel_trace.csv $O/bench_$lanes.json
done
