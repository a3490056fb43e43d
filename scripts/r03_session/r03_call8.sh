set -x
mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "masked" > gpurun_out/r03/pytest8.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03/pytest8.log
PROBE_PERM=1 timeout -k 10 300 python scripts/mconv_probe.py auto: waves8:BASIC_MCONV_DMA_WAVES=8 waves4:BASIC_MCONV_DMA_WAVES=4 > gpurun_out/r03/mconv_probe6.txt 2>&1; cat gpurun_out/r03/mconv_probe6.txt
timeout -k 10 300 python bench.py --workload checkerboard --no-cpu-baseline > gpurun_out/r03/bench_cb5.json 2> gpurun_out/r03/bench_cb5.err; head -c 1300 gpurun_out/r03/bench_cb5.json; tail -3 gpurun_out/r03/bench_cb5.err
