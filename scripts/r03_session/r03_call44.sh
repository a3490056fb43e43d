mkdir -p gpurun_out/r03
PROBE_SMALL_BATCHES=1 timeout -k 10 600 python scripts/scanline_probe.py > gpurun_out/r03/scanline_probe_small.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03/scanline_probe_small.txt | tail -5
