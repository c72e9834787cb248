#!/bin/bash
# Per-launch durations of column_waves_kernel (GPU box, via gpurun): rocprofv3 --kernel-trace over tools/pmc_column_waves.py, 60 launches
# of each job without the clock ramp, mean / median / min / max and the first 24 durations per job (profiles/r06_step4_table_in_device_memory.txt).
# MINARROW_HIP_LIB selects another build of the library to compare with. $@ = jobs (default: the four of the profile).
set -u
export TMPDIR=/tmp CW_NO_RAMP=1
O=$GRAFT_REPO_ROOT/gpurun_out/trace_cw; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
JOBS=${@:-columns_i32_gated columns_i32_dense chunks_as_one_i32_gated columns_i64_dense}
for job in $JOBS; do
  rm -rf $O/$job
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/$job -- python3 tools/pmc_column_waves.py 60 60000 $job > /dev/null 2>&1 || exit 1
  python3 - $O/$job <<'PY'
import csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
d = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(f)) if 'column_waves' in r['Kernel_Name'])
u = [(e - s) / 1e3 for s, e in d][2:]
print(sys.argv[1].split('/')[-1], len(u), f"mean {statistics.mean(u):.1f} median {statistics.median(u):.1f} min {min(u):.1f} max {max(u):.1f} |", ' '.join(f"{x:.0f}" for x in u[:24]))
PY
done
