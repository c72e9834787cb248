#!/bin/bash
# Runs on the GPU box (via gpurun): produces the round's evidence under gpurun_out/profiles/ — copy what should be
# judged into profiles/ afterwards. rocprofv3 gets the program itself after `--` (no wrappers), counters are
# collected in their own passes (no --pmc together with trace domains other than kernel-trace).
set -u
export TMPDIR=/tmp
R=${1:-r03}
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
python3 tools/pmc_summarize.py $O/${R}_pmc_fetch_counter_collection.csv $O/${R}_pmc_write_counter_collection.csv $R > $O/pmc_traffic.json
step "one process, group API, RCCL exchange (1 GPU)"; timeout -k 10 300 python3 bench.py --gpus 1 --force-group --no-cpu-baseline > $O/${R}_bench_group_1gpu_rccl.json 2> $O/${R}_bench_group.err || exit 1
step "launcher, one rank, native communicator"; timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --force-dist --no-cpu-baseline --no-other-configs > $O/${R}_bench_ranks_1gpu_native_comm.json 2> $O/${R}_bench_ranks.err || exit 1
step "group issue microbench (calling thread vs per-member issue threads)"; timeout -k 10 200 python3 tools/bench_group_issue.py > $O/${R}_group_issue.json 2> $O/${R}_group_issue.err || exit 1
step "the per-GPU share of the 8-way partition (125 M rows per column) on ONE GPU: what a strong-scaling step costs beyond its two scans"
L="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --force-dist --no-cpu-baseline --no-other-configs --steps 200 --warmup 10 --rows 125000000"
timeout -k 10 200 $L --overlap off > $O/${R}_strong_share_1gpu_ranks_overlap_off.json 2>/dev/null || exit 1
timeout -k 10 200 $L --overlap on > $O/${R}_strong_share_1gpu_ranks_overlap_on.json 2>/dev/null || exit 1
timeout -k 10 200 python3 bench.py --gpus 1 --force-group --no-cpu-baseline --no-other-configs --steps 200 --warmup 10 --rows 125000000 > $O/${R}_strong_share_1gpu_group_threads.json 2>/dev/null || exit 1
timeout -k 10 200 python3 bench.py --gpus 1 --force-group --group-issue caller --no-cpu-baseline --no-other-configs --steps 200 --warmup 10 --rows 125000000 > $O/${R}_strong_share_1gpu_group_caller.json 2>/dev/null || exit 1
timeout -k 10 200 python3 bench.py --gpus 1 --force-group --overlap on --no-cpu-baseline --no-other-configs --steps 200 --warmup 10 --rows 125000000 > $O/${R}_strong_share_1gpu_group_overlap_on.json 2>/dev/null || exit 1
timeout -k 10 200 $L --overlap on --exchange torch > $O/${R}_strong_share_1gpu_ranks_torch_overlap_on.json 2>/dev/null || exit 1
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-other-configs --steps 200 --warmup 10 --rows 125000000 > $O/${R}_strong_share_1gpu_plain.json 2>/dev/null || exit 1
step "Power series accuracy"; timeout -k 10 200 python3 tools/pow_series_report.py > $O/${R}_pow_series_accuracy.json 2>/dev/null || exit 1
step "kernel trace of the chunked regime (60 000 x 8192-row chunk pairs): kernel durations apart from table delivery"
MA_MATRIX_ONLY_SMALL_CHUNKS=1 timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 tools/bench_matrix.py --only super_array,consolidate --reps 3 > /dev/null 2>&1 || exit 1
python3 tools/trace_summary.py $O/trace kernel > $O/${R}_super_array_trace_final.txt; rm -rf $O/trace
step "matrix (on the runtime bench.py runs on)"; MA_IMPORT_TORCH=1 timeout -k 10 600 python3 tools/bench_matrix.py > $O/${R}_matrix.jsonl 2> $O/${R}_matrix.err || exit 1
step "size sweep"; MA_IMPORT_TORCH=1 timeout -k 10 400 python3 tools/sweep_sizes.py > $O/${R}_sweep_sizes.jsonl 2>/dev/null || exit 1
step "lanes"; timeout -k 10 200 python3 tools/bench_lanes.py > $O/${R}_lanes.json 2>/dev/null || exit 1
step "record-batch streams: ingestion and the stream operator at chunk-sized and large batches"
for a in "8192 20000" "65536 2000" "262144 500" "1000000 128" "8000000 32"; do timeout -k 10 200 python3 tools/bench_stream_ingest.py $a 2>/dev/null | tail -1; done > $O/${R}_stream_ingest.jsonl
for a in "8192 5000" "65536 1000" "262144 250" "1000000 64"; do timeout -k 10 200 python3 tools/bench_stream_op.py $a 2>/dev/null | tail -1; done > $O/${R}_stream_op.jsonl
step "chunk-list sums: the shipped wave kernel against the round's first shape, and pieces against segments"
timeout -k 10 300 python3 tools/probe_sum_chunks.py 0,4096 0,1,2,3 > $O/${R}_sweep_sum_chunks.jsonl 2>/dev/null || exit 1
timeout -k 10 200 python3 tools/probe_sum_chunks.py 0,4096 0 u8,i16 1000x4294912 > $O/${R}_sweep_sum_pieces.jsonl 2>/dev/null || exit 1
step "simd_eq_mask shapes"; timeout -k 10 200 python3 tools/sweep_eq_mask.py > $O/${R}_sweep_eq_mask.jsonl 2>/dev/null || exit 1
rm -rf $O/stats $O/pmc_fetch $O/pmc_write
ls -la $O
head -c 1200 $O/${R}_bench.json
