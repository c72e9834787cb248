#!/bin/bash
# Runs on the GPU box (via gpurun): produces the round's evidence under gpurun_out/profiles/ — copy what should be
# judged into profiles/ afterwards. rocprofv3 gets the program itself after `--` (no wrappers), counters are
# collected in their own passes (no --pmc together with trace domains other than kernel-trace).
set -u
export TMPDIR=/tmp
R=${1:-r02}
O=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $O
cd $GRAFT_REPO_ROOT
step() { echo "== $1"; }
step "bench (the driver's command)"; timeout -k 10 500 python3 bench.py --steps 20 --warmup 3 > $O/${R}_bench.json 2> $O/${R}_bench.err || exit 1
step "rocprof stats of the same command (kernel durations: sums AND the configs 3-5 kernels)"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/${R}_bench_under_rocprof.json 2>/dev/null || exit 1
cp $O/stats/*/*_kernel_stats.csv $O/${R}_bench_kernel_stats.csv 2>/dev/null
step "pmc fetch"; timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null 2>&1 || exit 1
step "pmc write"; timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null 2>&1 || exit 1
cp $O/pmc_fetch/*/*_counter_collection.csv $O/${R}_pmc_fetch_counter_collection.csv 2>/dev/null
cp $O/pmc_write/*/*_counter_collection.csv $O/${R}_pmc_write_counter_collection.csv 2>/dev/null
step "one process, group API, RCCL exchange (1 GPU)"; timeout -k 10 300 python3 bench.py --gpus 1 --force-group --no-cpu-baseline > $O/${R}_bench_group_1gpu_rccl.json 2> $O/${R}_bench_group.err || exit 1
step "launcher, one rank, native communicator"; timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --force-dist --no-cpu-baseline --no-other-configs > $O/${R}_bench_ranks_1gpu_native_comm.json 2> $O/${R}_bench_ranks.err || exit 1
step "matrix (on the runtime bench.py runs on)"; MA_IMPORT_TORCH=1 timeout -k 10 600 python3 tools/bench_matrix.py > $O/${R}_matrix.jsonl 2> $O/${R}_matrix.err || exit 1
step "size sweep"; MA_IMPORT_TORCH=1 timeout -k 10 400 python3 tools/sweep_sizes.py > $O/${R}_sweep_sizes.jsonl 2>/dev/null || exit 1
step "lanes"; timeout -k 10 200 python3 tools/bench_lanes.py > $O/${R}_lanes.json 2>/dev/null || exit 1
rm -rf $O/stats $O/pmc_fetch $O/pmc_write
ls -la $O
head -c 1200 $O/${R}_bench.json
