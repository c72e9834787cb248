#!/bin/bash
# Write-side counters of the library's own read + write kernels (GPU box, via gpurun): one pass for the durations, one for the
# counters (no trace domain beside kernel-trace). $1 = tag. Output: gpurun_out/profiles/<tag>_pmc_write_lib_*.csv and the summary
# <tag>_pmc_write_summary.txt.
set -u
export TMPDIR=/tmp
R=${1:-r06}
O=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 tools/pmc_write_lib.py 10 > $O/${R}_pmc_write_lib_wall.json || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pw_stats -- python3 tools/pmc_write_lib.py 6 > /dev/null 2>&1 || exit 1
cp $O/pw_stats/*/*_kernel_stats.csv $O/${R}_pmc_write_lib_kernel_stats.csv; rm -rf $O/pw_stats
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/pw_pmc -- python3 tools/pmc_write_lib.py 3 > $O/${R}_pmc_write_lib.out 2>&1 || { tail -5 $O/${R}_pmc_write_lib.out; exit 1; }
cp $O/pw_pmc/*/*_counter_collection.csv $O/${R}_pmc_write_lib_counters.csv; rm -rf $O/pw_pmc
python3 tools/pmc_write_lib_summary.py "$R" "$O" > $O/${R}_pmc_write_summary.txt
cat $O/${R}_pmc_write_summary.txt
