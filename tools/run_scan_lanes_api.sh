#!/bin/bash
# tools/bench_scan_lanes.py over the fused pair and the single-column kernels of four types (3-s pauses between the processes: the
# driver clears the VRAM its predecessor released in the background) -> profiles/r05_scan_lanes_api.jsonl
cd "$(dirname "$0")/.."
echo '# tools/run_scan_lanes_api.sh (round 5): back-to-back sums on one GPU, one stream against ma_scan_lanes_* (two streams, each scan started by the early stamp of the one before); four distinct columns (pairs) in turn; per scan: wall microseconds and TB/s of the bytes scanned'
echo '# i64 + f64 per scan: ma_sum_fused on an async context / ma_scan_lanes_sum_fused'
sleep 3; timeout -k 10 300 python3 tools/bench_scan_lanes.py 65536 262144 1048576 4194304 16777216 33554432 67108864 125000000 250000000 500000000
for t in l i C f; do
  echo "# one column per scan, format $t: ma_<t>_sum (enqueue-style) / ma_scan_lanes_sum"
  sleep 3; MA_BENCH_SINGLE=$t timeout -k 10 200 python3 tools/bench_scan_lanes.py 1048576 4194304 16777216 67108864 268435456 1000000000
done
