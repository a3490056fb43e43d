mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_gpu_scanline.py -q -m gpu -x > gpurun_out/r03/pytest21.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 gpurun_out/r03/pytest21.log
[ $rc -eq 0 ] && BASIC_SCAN_PROFILE=1 PROBE_SHAPES=1 timeout -k 10 300 python scripts/scanline_probe.py > gpurun_out/r03/scanline_profile2.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03/scanline_profile2.txt | tail -3
[ $rc -eq 0 ] && timeout -k 10 600 python -m pytest tests/test_gpu_ar_codecs.py tests/test_gpu_codec_graph.py tests/test_gpu_harness_workers.py -q -m gpu -x > gpurun_out/r03/pytest21b.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r03/pytest21b.log
[ $rc -eq 0 ] && KODAK_CFGS="basic:0 basic:3 basic:6" bash scripts/kodak_workers.sh gpurun_out/r03 > gpurun_out/r03/kodak_workers2.log 2>&1; cat gpurun_out/r03/kodak_workers/summary.txt
