mkdir -p gpurun_out/r03
timeout -k 10 600 python scripts/scanline_probe.py > gpurun_out/r03/scanline_probe_full.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03/scanline_probe_full.txt | tail -5
timeout -k 10 900 python -m pytest tests/test_gpu_scanline.py tests/test_gpu_ar_codecs.py tests/test_gpu_codec_graph.py tests/test_gpu_harness_workers.py tests/test_gpu_pgm.py -q -m gpu -x > gpurun_out/r03/pytest30.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r03/pytest30.log
