mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_gpu_scanline.py -q -m gpu -x -s > gpurun_out/r03/pytest24.log 2>&1; rc=$?; echo "pytest rc $rc"; grep -v amdgpu.ids gpurun_out/r03/pytest24.log | tail -25
[ $rc -eq 0 ] && BASIC_SCAN_PROFILE=1 PROBE_SHAPES=1 timeout -k 10 300 python scripts/scanline_probe.py > gpurun_out/r03/scanline_profile4.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03/scanline_profile4.txt | tail -3
[ $rc -eq 0 ] && KODAK_CFGS="basic:0 basic:4" bash scripts/kodak_workers.sh gpurun_out/r03 > gpurun_out/r03/kodak_workers4.log 2>&1; cat gpurun_out/r03/kodak_workers/summary.txt
