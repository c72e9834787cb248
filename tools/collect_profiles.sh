#!/bin/bash
# Runs on the GPU box (via gpurun): produces the round's evidence under gpurun_out/profiles/ — copy what should be judged into
# profiles/ afterwards. rocprofv3 gets the program itself after `--` (no wrappers), counters are collected in their own passes
# (no --pmc together with trace domains other than kernel-trace). A 3-s pause in front of every process: the driver clears the
# VRAM its predecessor released in the background, and scans read 5 % slower meanwhile (profiles/r04_read_rate_states_root_cause.txt).
set -u
export TMPDIR=/tmp
R=${1:-r05}
O=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $O
cd $GRAFT_REPO_ROOT
step() { echo "== $1"; sleep 3; }
step "bench (the driver's command)"; timeout -k 10 500 python3 bench.py --steps 20 --warmup 3 > $O/${R}_bench.json 2> $O/${R}_bench.err || exit 1
step "rocprof stats of the same command (kernel durations: sums AND the configs 3-5 kernels)"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-torch-hosted-leg > $O/${R}_bench_under_rocprof.json 2>/dev/null || exit 1
cp $O/stats/*/*_kernel_stats.csv $O/${R}_bench_kernel_stats.csv 2>/dev/null
step "pmc fetch (separate passes: the separate-launch step, then the fused one)"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-other-configs --no-torch-hosted-leg > /dev/null 2>&1 || exit 1
step "pmc write"; timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-other-configs --no-torch-hosted-leg > /dev/null 2>&1 || exit 1
step "pmc fetch, fused step"; timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_fused -- python3 bench.py --steps 3 --warmup 1 --ramp-ms 0 --step fused --no-cpu-baseline --no-other-configs --no-torch-hosted-leg > /dev/null 2>&1 || exit 1
step "pmc write, fused step"; timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_fused -- python3 bench.py --steps 3 --warmup 1 --ramp-ms 0 --step fused --no-cpu-baseline --no-other-configs --no-torch-hosted-leg > /dev/null 2>&1 || exit 1
{ head -1 $O/pmc_fetch/*/*_counter_collection.csv; tail -q -n +2 $O/pmc_fetch/*/*_counter_collection.csv $O/pmc_fetch_fused/*/*_counter_collection.csv; } > $O/${R}_pmc_fetch_counter_collection.csv
{ head -1 $O/pmc_write/*/*_counter_collection.csv; tail -q -n +2 $O/pmc_write/*/*_counter_collection.csv $O/pmc_write_fused/*/*_counter_collection.csv; } > $O/${R}_pmc_write_counter_collection.csv
python3 tools/pmc_summarize.py $O/${R}_pmc_fetch_counter_collection.csv $O/${R}_pmc_write_counter_collection.csv $R > $O/pmc_traffic.json
step "one process, group API, RCCL exchange (1 GPU)"; timeout -k 10 300 python3 bench.py --gpus 1 --force-group --no-cpu-baseline > $O/${R}_bench_group_1gpu_rccl.json 2> $O/${R}_bench_group.err || exit 1
step "launcher, one rank, native communicator, torch-free GPU path"; timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --force-dist --no-cpu-baseline > $O/${R}_bench_ranks_1gpu_native_comm.json 2> $O/${R}_bench_ranks.err || exit 1
step "the per-GPU share of the 8-way partition (125 M rows per column) on ONE GPU"; bash tools/run_share.sh gpurun_out/profiles/share > $O/${R}_share_1gpu.txt 2>&1
step "the sub-family kernels by counters"; bash tools/pmc_subfamily.sh $R > $O/${R}_subfamily.log 2>&1
step "size sweep (torch-free)"; timeout -k 10 400 python3 tools/sweep_sizes.py > $O/${R}_sweep_sizes.jsonl 2>/dev/null || exit 1
step "fused vs single-column sums"; timeout -k 10 300 python3 tools/sweep_fused.py > $O/${R}_sweep_fused.jsonl 2>/dev/null || exit 1
step "the chunked regime's forms on one block"; timeout -k 10 300 python3 tools/ab_chunked.py > $O/${R}_ab_chunked.jsonl 2>/dev/null || exit 1
step "the same on a searched (fast) output block"; MA_AB_SEARCH=1 timeout -k 10 300 python3 tools/ab_chunked.py > $O/${R}_ab_chunked_fast_block.jsonl 2>/dev/null || exit 1
step "matrix (torch-free)"; timeout -k 10 700 python3 tools/bench_matrix.py > $O/${R}_matrix.jsonl 2> $O/${R}_matrix.err || exit 1
step "record-batch streams"
for a in "8192 20000" "65536 2000" "1000000 128"; do timeout -k 10 200 python3 tools/bench_stream_ingest.py $a 2>/dev/null | tail -1; done > $O/${R}_stream_ingest.jsonl
for a in "8192 5000" "65536 1000" "1000000 64"; do timeout -k 10 200 python3 tools/bench_stream_op.py $a 2>/dev/null | tail -1; done > $O/${R}_stream_op.jsonl
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_fetch_fused $O/pmc_write_fused
ls -la $O
head -c 1500 $O/${R}_bench.json
