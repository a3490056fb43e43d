set -x
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_scanline.py tests/test_gpu_pgm.py tests/test_gpu_ar_codecs.py -q -m gpu -x > gpurun_out/r03/pytest12.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -6 gpurun_out/r03/pytest12.log
timeout -k 10 300 python scripts/scanline_probe.py > gpurun_out/r03/scanline_probe3.txt 2>&1; cat gpurun_out/r03/scanline_probe3.txt
BASIC_SCAN_DEBUG=2 PROBE_SHAPES=1 timeout -k 10 300 python scripts/scanline_probe.py > gpurun_out/r03/scanline_probe3_nostage.txt 2>&1; cat gpurun_out/r03/scanline_probe3_nostage.txt
BASIC_SCAN_DEBUG=4 PROBE_SHAPES=1 timeout -k 10 300 python scripts/scanline_probe.py > gpurun_out/r03/scanline_probe3_nodot.txt 2>&1; cat gpurun_out/r03/scanline_probe3_nodot.txt
