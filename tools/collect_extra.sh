#!/bin/bash
# Second evidence pass of a round (GPU box, via gpurun): PMC traffic of the configs 3-5 kernels, the matrix with its
# rocprofv3 kernel stats, a large random fuzz campaign. Outputs under gpurun_out/profiles/.
set -u
export TMPDIR=/tmp
R=${1:-r02}
O=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== pmc fetch (bench incl. configs 3-5)"; timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --other-reps 3 > /dev/null 2>&1 || exit 1
echo "== pmc write"; timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --other-reps 3 > /dev/null 2>&1 || exit 1
cp $O/pf/*/*_counter_collection.csv $O/${R}_pmc_fetch_all_configs.csv 2>/dev/null
cp $O/pw/*/*_counter_collection.csv $O/${R}_pmc_write_all_configs.csv 2>/dev/null
rm -rf $O/pf $O/pw
echo "== matrix"; MA_IMPORT_TORCH=1 timeout -k 10 600 python3 tools/bench_matrix.py > $O/${R}_matrix.jsonl 2> $O/${R}_matrix.err || exit 1
echo "== matrix under rocprof"; MA_IMPORT_TORCH=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sm -- python3 tools/bench_matrix.py --reps 2 > /dev/null 2>&1 || exit 1
cp $O/sm/*/*_kernel_stats.csv $O/${R}_matrix_kernel_stats.csv 2>/dev/null
rm -rf $O/sm
echo "== fuzz campaign"; MA_FUZZ_EXAMPLES=${FUZZ:-15000} MA_FUZZ_RANDOM=1 timeout -k 10 1000 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > $O/${R}_fuzz_campaign.log 2>&1; echo "rc=$?" >> $O/${R}_fuzz_campaign.log
tail -3 $O/${R}_fuzz_campaign.log
ls -la $O
