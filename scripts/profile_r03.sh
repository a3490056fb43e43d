#!/bin/bash
# round-3 profiling passes (run on the GPU box through gpurun); outputs under gpurun_out/r03prof/
set -e
export GIT_SHA=${GIT_SHA:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B1="--workers 1 --steps 2 --warmup 1 --no-cpu-baseline --no-dominant --no-extra-legs"
# 1. kernel trace + stats of the default bench (3 stream workers, whole batches in flight): per-kernel totals and the overlap timeline
rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs --no-ar-workloads > $O/bench_trace.log 2>&1
(cd $R && python scripts/prof_summary.py gpurun_out/r03prof/trace 24 > gpurun_out/r03prof/bench_summary.txt && python scripts/timeline.py gpurun_out/r03prof/trace 60 > gpurun_out/r03prof/bench_timeline.txt)
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
echo trace done
# 2. HBM traffic, separate passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 $R/bench.py $B1 > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 $R/bench.py $B1 > $O/write.log 2>&1
echo write done
(cd $R && python scripts/pmc_traffic.py gpurun_out/r03prof/fetch gpurun_out/r03prof/write gpurun_out/r03prof/r03_pmc_traffic.json ${GIT_SHA:-r03})
# 3. MFMA utilisation / LDS conflicts / MFMA op counts
rocprofv3 --kernel-trace --pmc MfmaUtil -d $O/mfma1 -o m --output-format csv -- python3 $R/bench.py $B1 > $O/mfma1.log 2>&1
echo mfma1 done
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/mfma2 -o m --output-format csv -- python3 $R/bench.py $B1 > $O/mfma2.log 2>&1
echo mfma2 done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/mfma3 -o m --output-format csv -- python3 $R/bench.py $B1 > $O/mfma3.log 2>&1
(cd $R && python scripts/pmc_mfma.py gpurun_out/r03prof/r03_pmc_mfma.json gpurun_out/r03prof/mfma1 gpurun_out/r03prof/mfma2 gpurun_out/r03prof/mfma3 > gpurun_out/r03prof/pmc_mfma.txt)
# 4. batch 32 (one rank's share of the strong leg): timeline of 6 workers, 4 token lanes
rocprofv3 --kernel-trace --output-format csv -d $O/trace_b32 -o b32 -- python3 $R/bench.py --batch 32 --workers 6 --token-lanes 4 --steps 96 --warmup 6 --no-cpu-baseline --no-extra-legs --no-dominant > $O/bench_b32.log 2>&1
(cd $R && python scripts/timeline.py gpurun_out/r03prof/trace_b32 16 > gpurun_out/r03prof/batch32_timeline.txt)
echo b32 done
# 5. the AR workloads: kernel stats
for wl in checkerboard basic; do
  rocprofv3 --kernel-trace --stats -d $O/ar_$wl -o ar --output-format csv -- python3 $R/bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline > $O/ar_$wl.log 2>&1
  cp $(find $O/ar_$wl -name "*kernel_stats.csv" | head -1) $O/r03_ar_kernel_stats_$wl.csv
  echo ar $wl done
done
# keep the merge small
rm -rf $O/trace $O/fetch $O/write $O/mfma1 $O/mfma2 $O/mfma3 $O/trace_b32 $O/ar_checkerboard $O/ar_basic
echo all done
