set -x
mkdir -p gpurun_out/r03
cd scripts/micro && ./mfma_arith > ../../gpurun_out/r03/mfma_arith.txt 2>&1; cd ../..
cat gpurun_out/r03/mfma_arith.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest1.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03/pytest1.log
timeout -k 10 300 python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; tail -c 600 gpurun_out/r03/bench_default.json
BATCH=32 STEPS=96 OUT=gpurun_out/r03/sweep_b32.txt SWEEP="3:steps:8 5:steps:8 6:steps:8 8:steps:8 10:steps:8 12:steps:8 6:steps:4 8:steps:4 10:steps:4 6:steps:1 8:steps:1 10:steps:1 8:steps:2" bash scripts/workers_sweep.sh
