mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_harness_workers.py tests/test_gpu_benchmark.py -q -m gpu -x > gpurun_out/r03/pytest41.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r03/pytest41.log
[ $rc -eq 0 ] || exit 1
KODAK_CFGS="basic:3 basic:4 basic:6 hyperprior:3" bash scripts/kodak_workers.sh gpurun_out/r03 > gpurun_out/r03/kodak_workers10.log 2>&1; cat gpurun_out/r03/kodak_workers/summary.txt
