mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_gpu_scanline.py -q -m gpu -x > gpurun_out/r03/pytest23.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 gpurun_out/r03/pytest23.log
[ $rc -eq 0 ] && BASIC_SCAN_PROFILE=1 PROBE_SHAPES=1 timeout -k 10 300 python scripts/scanline_probe.py > gpurun_out/r03/scanline_profile3.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03/scanline_profile3.txt | tail -4
[ $rc -eq 0 ] && KODAK_CFGS="basic:0 basic:4" bash scripts/kodak_workers.sh gpurun_out/r03 > gpurun_out/r03/kodak_workers3.log 2>&1; cat gpurun_out/r03/kodak_workers/summary.txt
