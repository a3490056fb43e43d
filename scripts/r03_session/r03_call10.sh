set -x
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r03/pytest10.log 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/r03/pytest10.log
( time timeout -k 10 600 python bench.py ) > gpurun_out/r03/bench_default3.json 2> gpurun_out/r03/bench_default3.err; tail -c 600 gpurun_out/r03/bench_default3.json; tail -5 gpurun_out/r03/bench_default3.err
