#!/bin/bash
# stall / scalar-cache counters of the first and last transform layer (separate --pmc passes, kernel trace only)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/edge
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B1="--workers 1 --steps 2 --warmup 1 --no-cpu-baseline --no-dominant --no-extra-legs"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SMEM" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/bench.py $B1 > $O/p$i.log 2>&1
  echo pass $i done
done
(cd $R && python scripts/pmc_fold.py gpurun_out/edge/edge_layers.json deconv5s2_cout3,conv5x5_cin4_gdn,conv_tap_mfma_kernel\<4,\ 4,\ 5 gpurun_out/edge/p1 gpurun_out/edge/p2 gpurun_out/edge/p3 gpurun_out/edge/p4 gpurun_out/edge/p5 > gpurun_out/edge/edge_layers.txt)
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
cat $R/gpurun_out/edge/edge_layers.txt
