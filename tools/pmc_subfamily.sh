#!/bin/bash
# Counter passes over tools/pmc_subfamily.py (GPU box, via gpurun). The program itself after `--`; counters in passes of their
# own (no trace domain but kernel-trace), as the guide prescribes. Output: gpurun_out/profiles/<round>_subfamily_*.csv.
set -u
export TMPDIR=/tmp
R=${1:-r05}
O=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $O
cd $GRAFT_REPO_ROOT
run() {  # name, counters...
    local name=$1; shift
    echo "== $name: $*"
    timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $O/sf_$name -- python3 tools/pmc_subfamily.py 3 > $O/${R}_subfamily_$name.out 2>&1 || { echo "pass $name failed"; tail -5 $O/${R}_subfamily_$name.out; return 1; }
    cp $O/sf_$name/*/*_counter_collection.csv $O/${R}_subfamily_$name.csv 2>/dev/null
    rm -rf $O/sf_$name
}
echo "== kernel durations"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sf_stats -- python3 tools/pmc_subfamily.py 5 > $O/${R}_subfamily_wall.json 2>/dev/null || exit 1
cp $O/sf_stats/*/*_kernel_stats.csv $O/${R}_subfamily_kernel_stats.csv 2>/dev/null
rm -rf $O/sf_stats
run cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES || exit 1
run insts SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM || exit 1
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
ls -la $O | grep subfamily
