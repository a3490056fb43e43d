set -x
mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_gpu_scanline.py -q -m gpu -x > gpurun_out/r03/pytest11.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -15 gpurun_out/r03/pytest11.log
[ $rc -eq 0 ] && timeout -k 10 300 python scripts/scanline_probe.py > gpurun_out/r03/scanline_probe2.txt 2>&1; cat gpurun_out/r03/scanline_probe2.txt
[ $rc -eq 0 ] && timeout -k 10 600 python -m pytest tests/test_gpu_ar_codecs.py tests/test_gpu_harness_workers.py tests/test_gpu_benchmark.py -q -m gpu -x > gpurun_out/r03/pytest11b.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03/pytest11b.log
