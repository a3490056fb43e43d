mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_gpu_scanline.py -q -m gpu -x -s > gpurun_out/r03/pytest24.log 2>&1; rc=$?; echo "pytest rc $rc"; grep -v "amdgpu.ids\|\[None\]\|\[generic\]" gpurun_out/r03/pytest24.log | tail -25
[ $rc -eq 0 ] || exit 1
BASIC_SCAN_PROFILE=1 PROBE_SHAPES=1 timeout -k 10 300 python scripts/scanline_probe.py > gpurun_out/r03/scanline_profile4.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03/scanline_profile4.txt | tail -3
KODAK_CFGS="basic:0 basic:3 basic:4" bash scripts/kodak_workers.sh gpurun_out/r03 > gpurun_out/r03/kodak_workers4.log 2>&1; cat gpurun_out/r03/kodak_workers/summary.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
for w in 4; do
  rm -rf /tmp/kt$w
  (cd $R && timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$w -- python3 tools/run_benchmark.py --warmup --codec basic --synthetic 12 --height 512 --width 768 --batch-size 1 --workers $w --complexity-levels 0 --out /tmp/kt_out$w > /tmp/kt$w.json 2> /tmp/kt$w.err) || { echo "trace w$w failed"; tail -5 /tmp/kt$w.err; }
  python3 $R/scripts/kodak_timeline.py /tmp/kt$w 1 > $R/gpurun_out/r03/kodak_timeline_w$w.txt 2>&1
  grep -A200 "persistent launches" $R/gpurun_out/r03/kodak_timeline_w$w.txt | head -150
done
