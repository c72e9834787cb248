#!/bin/bash
# Counter passes over tools/pmc_column_waves.py (GPU box, via gpurun): the program itself after `--`, counters in passes of their
# own. Output: gpurun_out/profiles/<tag>_column_waves_*.csv. $1 = tag (e.g. r06_before), $2 = columns (default 60000).
set -u
export TMPDIR=/tmp
R=${1:-r06}
K=${2:-60000}
O=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $O
cd $GRAFT_REPO_ROOT
run() {  # name, counters...
    local name=$1; shift
    echo "== $name: $*"
    timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $O/cw_$name -- python3 tools/pmc_column_waves.py 3 $K > $O/${R}_column_waves_$name.out 2>&1 || { echo "pass $name failed"; tail -5 $O/${R}_column_waves_$name.out; return 1; }
    cp $O/cw_$name/*/*_counter_collection.csv $O/${R}_column_waves_$name.csv 2>/dev/null
    rm -rf $O/cw_$name
}
echo "== wall (no profiler)"
timeout -k 10 300 python3 tools/pmc_column_waves.py 20 $K > $O/${R}_column_waves_wall.json || exit 1
cat $O/${R}_column_waves_wall.json
echo "== kernel durations"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cw_stats -- python3 tools/pmc_column_waves.py 5 $K > /dev/null 2>&1 || exit 1
cp $O/cw_stats/*/*_kernel_stats.csv $O/${R}_column_waves_kernel_stats.csv 2>/dev/null
rm -rf $O/cw_stats
run cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES || exit 1
run insts SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM || exit 1
run fetch FETCH_SIZE || exit 1
ls -la $O | grep column_waves
