#!/bin/bash
# round-2 profiling passes (run on the GPU box through gpurun); outputs under gpurun_out/r02prof/
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B1="--workers 1 --steps 2 --warmup 1 --no-cpu-baseline --no-dominant --no-extra-legs"
# 1. kernel trace + stats of the default bench (3 stream workers, whole batches in flight): per-kernel totals and the overlap timeline
rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs > $O/bench_trace.log 2>&1
(cd $R && python scripts/prof_summary.py gpurun_out/r02prof/trace 24 > gpurun_out/r02prof/bench_summary.txt && python scripts/timeline.py gpurun_out/r02prof/trace 60 > gpurun_out/r02prof/bench_timeline.txt)
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
echo trace done
# 2. HBM traffic, separate passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 $R/bench.py $B1 > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 $R/bench.py $B1 > $O/write.log 2>&1
echo write done
(cd $R && python scripts/pmc_traffic.py gpurun_out/r02prof/fetch gpurun_out/r02prof/write gpurun_out/r02prof/r02_pmc_traffic.json ${GIT_SHA:-r02})
# 3. MFMA utilisation / LDS conflicts / MFMA op counts
rocprofv3 --kernel-trace --pmc MfmaUtil -d $O/mfma1 -o m --output-format csv -- python3 $R/bench.py $B1 > $O/mfma1.log 2>&1
echo mfma1 done
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/mfma2 -o m --output-format csv -- python3 $R/bench.py $B1 > $O/mfma2.log 2>&1
echo mfma2 done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/mfma3 -o m --output-format csv -- python3 $R/bench.py $B1 > $O/mfma3.log 2>&1
(cd $R && python scripts/pmc_mfma.py gpurun_out/r02prof/r02_pmc_mfma.json gpurun_out/r02prof/mfma1 gpurun_out/r02prof/mfma2 gpurun_out/r02prof/mfma3 > gpurun_out/r02prof/pmc_mfma.txt)
# keep the merge small
rm -rf $O/trace $O/fetch $O/write $O/mfma1 $O/mfma2 $O/mfma3
echo all done
