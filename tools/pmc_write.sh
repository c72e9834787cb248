#!/bin/bash
# On the GPU box: tools/pmc_write.hip under rocprofv3, one counter set per pass (kernel-trace for the durations).
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_write
mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $GRAFT_REPO_ROOT/tools/pmc_write.hip -o /tmp/pmc_write || exit 1
cd /tmp
timeout -k 10 120 /tmp/pmc_write 8 > $O/plain.txt 2>&1 || exit 1
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_sum" \
           "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_64B_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pass$i -- /tmp/pmc_write 8 > $O/pass$i.txt 2>&1 || { echo "pass $i failed"; tail -5 $O/pass$i.txt; exit 1; }
  cp $O/pass$i/*/*_counter_collection.csv $O/pass${i}_counters.csv 2>/dev/null
  cp $O/pass$i/*/*_kernel_trace.csv $O/pass${i}_trace.csv 2>/dev/null
  rm -rf $O/pass$i
done
ls -la $O
cat $O/plain.txt
