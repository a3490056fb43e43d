#!/bin/bash
# cache / stall counters of the masked-convolution launches of the checkerboard workload (separate --pmc passes, kernel trace only)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B1="--workload checkerboard --workers 1 --steps 1 --warmup 1 --no-cpu-baseline"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_LEVEL_VMEM TCP_TCC_READ_REQ_LATENCY TCP_TCP_TA_DATA_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/bench.py $B1 > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/p$i.log; }
  echo pass $i done
done
(cd $R && python scripts/pmc_fold.py gpurun_out/mc/mconv.json masked_conv_pos_kernel gpurun_out/mc/p1 gpurun_out/mc/p2 gpurun_out/mc/p3 gpurun_out/mc/p4 gpurun_out/mc/p5 > gpurun_out/mc/mconv.txt)
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
cat $R/gpurun_out/mc/mconv.txt
