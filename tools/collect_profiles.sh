#!/bin/bash
# Runs on the GPU box (via gpurun): produces the round's evidence under gpurun_out/profiles/ — copy what should be judged into
# profiles/ afterwards. rocprofv3 gets the program itself after `--` (no wrappers), counters are collected in their own passes
# (no --pmc together with trace domains other than kernel-trace). A 3-s pause in front of every process: the driver clears the
# VRAM its predecessor released in the background, and scans read 5 % slower meanwhile (profiles/r04_read_rate_states_root_cause.txt).
set -u
export TMPDIR=/tmp
R=${1:-r06}
PART=${2:-all}   # a | b | all: a gpurun call is limited to 20 minutes — part a = the bench lines and counters, part b = sweeps and the matrix
O=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $O
cd $GRAFT_REPO_ROOT
step() { echo "== $1"; sleep 3; }
if [ "$PART" != "b" ]; then
step "bench (the driver's command)"; timeout -k 10 500 python3 bench.py --steps 20 --warmup 3 > $O/${R}_bench.json 2> $O/${R}_bench.err || exit 1
step "rocprof stats of the same command (kernel durations: sums AND the configs 3-5 kernels)"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/${R}_bench_under_rocprof.json 2>/dev/null || exit 1
cp $O/stats/*/*_kernel_stats.csv $O/${R}_bench_kernel_stats.csv 2>/dev/null
step "pmc fetch (separate passes: the separate-launch step, then the fused one)"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-other-configs > /dev/null 2>&1 || exit 1
step "pmc write"; timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-other-configs > /dev/null 2>&1 || exit 1
step "pmc fetch, fused step"; timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_fused -- python3 bench.py --steps 3 --warmup 1 --ramp-ms 0 --step fused --no-cpu-baseline --no-other-configs > /dev/null 2>&1 || exit 1
step "pmc write, fused step"; timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_fused -- python3 bench.py --steps 3 --warmup 1 --ramp-ms 0 --step fused --no-cpu-baseline --no-other-configs > /dev/null 2>&1 || exit 1
{ head -1 $O/pmc_fetch/*/*_counter_collection.csv; tail -q -n +2 $O/pmc_fetch/*/*_counter_collection.csv $O/pmc_fetch_fused/*/*_counter_collection.csv; } > $O/${R}_pmc_fetch_counter_collection.csv
{ head -1 $O/pmc_write/*/*_counter_collection.csv; tail -q -n +2 $O/pmc_write/*/*_counter_collection.csv $O/pmc_write_fused/*/*_counter_collection.csv; } > $O/${R}_pmc_write_counter_collection.csv
python3 tools/pmc_summarize.py $O/${R}_pmc_fetch_counter_collection.csv $O/${R}_pmc_write_counter_collection.csv $R > $O/pmc_traffic.json
step "one process, group API, RCCL exchange (1 GPU)"; timeout -k 10 300 python3 bench.py --gpus 1 --force-group --no-cpu-baseline > $O/${R}_bench_group_1gpu_rccl.json 2> $O/${R}_bench_group.err || exit 1
step "launcher, one rank, native communicator, torch-free GPU path"; timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --force-dist --no-cpu-baseline > $O/${R}_bench_ranks_1gpu_native_comm.json 2> $O/${R}_bench_ranks.err || exit 1
step "REHEARSAL (loopback collective double): 8 members in one process, then 2 rank processes, on this one GPU"
DBL="MINARROW_HIP_RCCL_PATH=$GRAFT_REPO_ROOT/tests/loopback_rccl/libloopback_rccl.so"
env $DBL GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python3 bench.py --gpus 8 --rows 1000000000 --scan-lanes on --no-cpu-baseline --no-other-configs > $O/${R}_bench_rehearsal_group_8_members_one_gpu.json 2> $O/${R}_bench_rehearsal_group.err
env $DBL timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --rows 1000000000 --scan-lanes on --no-cpu-baseline --no-other-configs > $O/${R}_bench_rehearsal_ranks_2_processes_one_gpu.json 2> $O/${R}_bench_rehearsal_ranks.err
step "the 8192-row chunk regime by counters (60 000 columns, then 10^9 rows)"; bash tools/pmc_column_waves.sh $R 60000 > $O/${R}_column_waves.log 2>&1
timeout -k 10 200 python3 tools/pmc_column_waves.py 20 122070 > $O/${R}_column_waves_wall_1e9_rows.json 2>/dev/null
step "write-side counters of the library's read + write kernels"; bash tools/pmc_write_lib.sh $R > $O/${R}_pmc_write_lib.log 2>&1
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_fetch_fused $O/pmc_write_fused
fi
if [ "$PART" = "a" ]; then ls -la $O; head -c 1500 $O/${R}_bench.json; exit 0; fi
step "the sub-family kernels by counters"; bash tools/pmc_subfamily.sh $R > $O/${R}_subfamily.log 2>&1
step "size sweep (torch-free)"; timeout -k 10 400 python3 tools/sweep_sizes.py > $O/${R}_sweep_sizes.jsonl 2>/dev/null || exit 1
step "matrix (torch-free)"; timeout -k 10 700 python3 tools/bench_matrix.py > $O/${R}_matrix.jsonl 2> $O/${R}_matrix.err || exit 1
step "record-batch streams"
for a in "8192 20000" "65536 2000" "1000000 128"; do timeout -k 10 200 python3 tools/bench_stream_ingest.py $a 2>/dev/null | tail -1; done > $O/${R}_stream_ingest.jsonl
for a in "8192 5000" "65536 1000" "1000000 64"; do timeout -k 10 200 python3 tools/bench_stream_op.py $a 2>/dev/null | tail -1; done > $O/${R}_stream_op.jsonl
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_fetch_fused $O/pmc_write_fused
ls -la $O
[ -f $O/${R}_bench.json ] && head -c 1500 $O/${R}_bench.json || true
