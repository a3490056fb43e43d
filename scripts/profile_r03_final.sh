#!/bin/bash
# end-of-round-3 profiling passes (through gpurun); outputs under gpurun_out/r03final/ -- kernel trace + stats of the default bench,
# kernel stats of the AR workloads, kernel traces of the Kodak-shaped BaSIC harness run with 0 and 3 stream workers
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs --no-ar-workloads > $O/bench_trace.log 2>&1 || echo "bench trace failed"
(cd $R && python scripts/prof_summary.py gpurun_out/r03final/trace 24 > gpurun_out/r03final/bench_summary.txt && python scripts/timeline.py gpurun_out/r03final/trace 60 > gpurun_out/r03final/bench_timeline.txt)
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
echo trace done
for wl in checkerboard basic; do
  rocprofv3 --kernel-trace --stats -d $O/ar_$wl -o ar --output-format csv -- python3 $R/bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline > $O/ar_$wl.log 2>&1 || echo "ar $wl failed"
  cp $(find $O/ar_$wl -name "*kernel_stats.csv" | head -1) $O/ar_kernel_stats_$wl.csv
  echo ar $wl done
done
for w in 0 3; do
  (cd $R && timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$w -- python3 tools/run_benchmark.py --warmup --codec basic --synthetic 24 --height 512 --width 768 --batch-size 1 --workers $w --complexity-levels 0 --out /tmp/kt_out$w > /tmp/kt$w.json 2> /tmp/kt$w.err) || { echo "kodak trace w$w failed"; tail -5 /tmp/kt$w.err; }
  python3 $R/scripts/kodak_timeline.py /tmp/kt$w 1 > $O/kodak_timeline_w$w.txt 2>&1
  echo kodak w$w done
done
rm -rf $O/trace $O/ar_checkerboard $O/ar_basic
head -12 $O/kodak_timeline_w0.txt
head -16 $O/bench_summary.txt
