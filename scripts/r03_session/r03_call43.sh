mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_scanline.py -q -m gpu -x -s > gpurun_out/r03/pytest43.log 2>&1; rc=$?; echo "pytest rc $rc"; grep "pipelined\]\|passed\|failed" gpurun_out/r03/pytest43.log | tail -12
