set -x
mkdir -p gpurun_out/r03
timeout -k 10 300 python scripts/mconv_probe.py default: nostage:BASIC_MCONV_DEBUG=1 nostore:BASIC_MCONV_DEBUG=4 nostage_nostore:BASIC_MCONV_DEBUG=5 > gpurun_out/r03/mconv_probe2.txt 2>&1; cat gpurun_out/r03/mconv_probe2.txt
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r03/pytest4.log 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/r03/pytest4.log
timeout -k 10 300 python bench.py --workload checkerboard --no-cpu-baseline > gpurun_out/r03/bench_cb2.json 2> gpurun_out/r03/bench_cb2.err; head -c 900 gpurun_out/r03/bench_cb2.json
