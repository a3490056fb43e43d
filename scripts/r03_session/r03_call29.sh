mkdir -p gpurun_out/r03
BASIC_SCAN_PROFILE=1 PROBE_SHAPES=1 PROBE_PERSISTENT_ONLY=1 timeout -k 10 300 python scripts/scanline_probe.py > gpurun_out/r03/scanline_profile5.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03/scanline_profile5.txt | tail -4
KODAK_CFGS="basic:0" bash scripts/kodak_workers.sh gpurun_out/r03 > gpurun_out/r03/kodak_workers5.log 2>&1; cat gpurun_out/r03/kodak_workers/summary.txt
