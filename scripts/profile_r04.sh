#!/bin/bash
# round-4 profiling passes (through gpurun); outputs under gpurun_out/r04prof/.  STEP=1..5 selects a part (each within a gpurun call's limit)
export GIT_SHA=${GIT_SHA:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B1="--workers 1 --steps 2 --warmup 1 --no-cpu-baseline --no-dominant --no-extra-legs"
AR="--workers 1 --steps 1 --warmup 1 --levels 0 --no-kodak-leg --no-cpu-baseline"
if [ "${STEP:-1}" = 1 ]; then
  # kernel trace + stats of the default bench (probed workers, whole batches in flight): per-kernel totals and the overlap timeline
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs --no-ar-workloads > $O/bench_trace.log 2>&1 || echo "bench trace failed"
  (cd $R && python scripts/prof_summary.py gpurun_out/r04prof/trace 24 > gpurun_out/r04prof/bench_summary.txt && python scripts/timeline.py gpurun_out/r04prof/trace 60 > gpurun_out/r04prof/bench_timeline.txt)
  cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/r04_bench_kernel_stats.csv
  rm -rf $O/trace
  echo trace done; head -14 $O/bench_summary.txt
fi
if [ "${STEP:-1}" = 2 ]; then
  # HBM traffic of the transforms, separate passes
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 $R/bench.py $B1 > $O/fetch.log 2>&1 || echo "fetch failed"
  echo fetch done
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 $R/bench.py $B1 > $O/write.log 2>&1 || echo "write failed"
  echo write done
  (cd $R && python scripts/pmc_traffic.py gpurun_out/r04prof/fetch gpurun_out/r04prof/write gpurun_out/r04prof/r04_pmc_traffic.json $GIT_SHA)
  rm -rf $O/fetch $O/write
fi
if [ "${STEP:-1}" = 3 ]; then
  # AR workloads: L2 -> fabric request counters of the dominant kernels (FETCH_SIZE / WRITE_SIZE abort on them), separate passes
  for wl in basic checkerboard; do
    i=0
    for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
      i=$((i+1))
      timeout -k 10 250 rocprofv3 --kernel-trace --pmc $set -d $O/ar_${wl}_$i -o p --output-format csv -- python3 $R/bench.py --workload $wl $AR > $O/ar_${wl}_$i.log 2>&1 || { echo "ar $wl pass $i failed"; tail -3 $O/ar_${wl}_$i.log; }
      echo ar $wl pass $i done
    done
  done
  (cd $R && python scripts/pmc_ar_fold.py gpurun_out/r04prof/r04_pmc_ar.json $GIT_SHA basic=gpurun_out/r04prof/ar_basic_1,gpurun_out/r04prof/ar_basic_2 checkerboard=gpurun_out/r04prof/ar_checkerboard_1,gpurun_out/r04prof/ar_checkerboard_2)
  rm -rf $O/ar_basic_1 $O/ar_basic_2 $O/ar_checkerboard_1 $O/ar_checkerboard_2
fi
if [ "${STEP:-1}" = 4 ]; then
  # AR workloads: kernel stats (6 workers) + matrix-core utilisation of the scan kernel
  for wl in checkerboard basic; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ars_$wl -o ar --output-format csv -- python3 $R/bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-kodak-leg --levels 0 > $O/ars_$wl.log 2>&1 || echo "ar stats $wl failed"
    cp $(find $O/ars_$wl -name "*kernel_stats.csv" | head -1) $O/r04_ar_kernel_stats_$wl.csv
    rm -rf $O/ars_$wl
    echo ar stats $wl done
  done
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc MfmaUtil SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/mfma_scan -o m --output-format csv -- python3 $R/bench.py --workload basic $AR > $O/mfma_scan.log 2>&1 || echo "mfma scan failed"
  (cd $R && python scripts/pmc_fold.py gpurun_out/r04prof/r04_pmc_scan_mfma.json scanline_batched gpurun_out/r04prof/mfma_scan > gpurun_out/r04prof/r04_pmc_scan_mfma.txt)
  rm -rf $O/mfma_scan
  cat $O/r04_pmc_scan_mfma.txt
fi
if [ "${STEP:-1}" = 5 ]; then
  # the batched scan-line kernel alone: launch times, in-kernel step profile, per-step path beside it
  cd $R
  timeout -k 10 200 python scripts/scanline_probe.py > $O/scanline_probe.txt 2>&1
  BASIC_SCAN_PROFILE=1 PROBE=64x16x16,8x16x16,24x32x48 timeout -k 10 200 python scripts/scan_batched_probe.py > $O/scan_batched_probe.txt 2>&1
  grep -v "^/opt" $O/scanline_probe.txt; grep "^B=" $O/scan_batched_probe.txt
fi
echo step ${STEP:-1} done
