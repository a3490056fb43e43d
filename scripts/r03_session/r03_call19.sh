mkdir -p gpurun_out/r03
BASIC_SCAN_PROFILE=1 PROBE_SHAPES=1 timeout -k 10 300 python scripts/scanline_probe.py > gpurun_out/r03/scanline_profile.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03/scanline_profile.txt | tail -12
