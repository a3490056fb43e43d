mkdir -p gpurun_out/r03
R=$PWD
cd /tmp && export TMPDIR=/tmp
for w in 0 6; do
  rm -rf /tmp/kt$w
  (cd $R && timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$w -- python3 tools/run_benchmark.py --warmup --codec basic --synthetic 12 --height 512 --width 768 --batch-size 1 --workers $w --complexity-levels 0 --out /tmp/kt_out$w > /tmp/kt$w.json 2> /tmp/kt$w.err) || { echo "trace w$w failed"; tail -5 /tmp/kt$w.err; }
  python3 $R/scripts/kodak_timeline.py /tmp/kt$w 170 > $R/gpurun_out/r03/kodak_timeline_w$w.txt 2>&1
  head -40 $R/gpurun_out/r03/kodak_timeline_w$w.txt
done
