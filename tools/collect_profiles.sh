#!/bin/bash
# Runs on the GPU box (via gpurun): produces the round's evidence under gpurun_out/profiles/ — copy what should be
# judged into profiles/ afterwards. rocprofv3 gets the program itself after `--` (no wrappers), counters are
# collected in their own passes (no --pmc together with trace domains other than kernel-trace).
set -u
export TMPDIR=/tmp
R=${1:-r01}
O=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== bench" ; timeout 600 python3 bench.py --steps 20 --warmup 3 > $O/${R}_bench.json 2> $O/${R}_bench.err ; echo rc=$?
echo "== rocprof stats" ; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/${R}_bench_under_rocprof.json 2>/dev/null ; echo rc=$?
cp $O/stats/*/*_kernel_stats.csv $O/${R}_bench_kernel_stats.csv 2>/dev/null
echo "== pmc fetch" ; timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 ; echo rc=$?
echo "== pmc write" ; timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 ; echo rc=$?
cp $O/pmc_fetch/*/*_counter_collection.csv $O/${R}_pmc_fetch_counter_collection.csv 2>/dev/null
cp $O/pmc_write/*/*_counter_collection.csv $O/${R}_pmc_write_counter_collection.csv 2>/dev/null
echo "== configs" ; timeout 900 python3 tools/bench_configs.py --configs 3,4,5,x > $O/${R}_configs.jsonl 2>/dev/null ; echo rc=$?
echo "== configs under rocprof" ; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg -- python3 tools/bench_configs.py --configs 3,4,5,x > /dev/null 2>&1 ; echo rc=$?
cp $O/stats_cfg/*/*_kernel_stats.csv $O/${R}_configs_kernel_stats.csv 2>/dev/null
echo "== sweeps" ; timeout 900 python3 tools/sweep_sum.py --types i64,f64 --variants 0,16,6,22,4 --bpcs 0,1,2 --rounds 3 --reps 10 > $O/${R}_sweep_sum.txt 2>&1 ; echo rc=$?
hipcc -O3 --offload-arch=gfx950 tools/ubench_sum.hip -o /tmp/ubench_sum 2>/dev/null && timeout 600 /tmp/ubench_sum 1000000000 3 > $O/${R}_ubench_sum.txt 2>&1
echo "== matrix" ; timeout 900 python3 tools/bench_matrix.py > $O/${R}_matrix.jsonl 2> $O/${R}_matrix.err ; echo rc=$?
echo "== matrix under rocprof" ; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mx -- python3 tools/bench_matrix.py --reps 2 > /dev/null 2>&1 ; echo rc=$?
cp $O/stats_mx/*/*_kernel_stats.csv $O/${R}_matrix_kernel_stats.csv 2>/dev/null
echo "== config 1 / 5 and the full-size config 5 (8 x 10^9-row batches, one column at a time: ~130 GB of HBM)"
timeout 600 python3 tools/bench_configs.py --configs 1,5 > $O/${R}_configs_1_5.jsonl 2>/dev/null ; echo rc=$?
timeout 600 python3 tools/bench_configs.py --configs 5 --config5-rows 1000000000 > $O/${R}_config5_full_size_1gpu.jsonl 2>/dev/null ; echo rc=$?
echo "== size sweep" ; timeout 600 python3 tools/sweep_sizes.py > $O/${R}_sweep_sizes.jsonl 2>/dev/null ; echo rc=$?
echo "== host-resident operands (PCIe-inclusive, never the headline)" ; timeout 600 python3 tools/bench_pcie.py > $O/${R}_pcie_tiled.json 2>/dev/null ; echo rc=$?
echo "== launch-bound shapes" ; timeout 600 python3 tools/bench_graph.py > $O/${R}_launch_bound.json 2>/dev/null ; echo rc=$?
rm -rf $O/stats $O/stats_cfg $O/stats_mx $O/pmc_fetch $O/pmc_write
ls -la $O
head -c 1500 $O/${R}_bench.json
