#!/bin/bash
# cache / stall / MFMA counters of masked_conv_dma_kernel on the 1536 -> 1536 merger layer at 32,768 checkerboard positions
# (scripts/mconv_probe.py, step-contiguous planes): separate --pmc passes, kernel trace only
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mc3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PROBE_PERM=1 PROBE_LAYERS=${PROBE_LAYERS:-1}
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "MfmaUtil" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "FETCH_SIZE WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/scripts/mconv_probe.py perm: > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/p$i.log; }
  echo pass $i done
done
(cd $R && python scripts/pmc_fold.py gpurun_out/mc3/mconv.json masked_conv gpurun_out/mc3/p1 gpurun_out/mc3/p2 gpurun_out/mc3/p3 gpurun_out/mc3/p4 gpurun_out/mc3/p5 gpurun_out/mc3/p6 gpurun_out/mc3/p7 gpurun_out/mc3/p8 > gpurun_out/mc3/mconv.txt)
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6 $O/p7 $O/p8
cat $R/gpurun_out/mc3/mconv.txt
