set -x
mkdir -p gpurun_out/r03
(cd scripts/micro && ./blds_check) > gpurun_out/r03/blds_check.txt 2>&1; cat gpurun_out/r03/blds_check.txt
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_codec_graph.py tests/test_gpu_reference_kats.py tests/test_gpu_pgm.py tests/test_gpu_ar_codecs.py tests/test_gpu_scanline.py -q -m gpu -x > gpurun_out/r03/pytest2.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r03/pytest2.log
timeout -k 10 300 python bench.py --workload checkerboard --no-cpu-baseline > gpurun_out/r03/bench_cb.json 2> gpurun_out/r03/bench_cb.err; tail -c 1500 gpurun_out/r03/bench_cb.json; tail -3 gpurun_out/r03/bench_cb.err
BATCH=32 STEPS=96 OUT=gpurun_out/r03/sweep_b32.txt SWEEP="3:steps:8 4:steps:8 5:steps:8 6:steps:8 7:steps:8 6:steps:4 6:steps:2 8:steps:8:16 8:steps:8:24 10:steps:8:24 6:steps:8:16" bash scripts/workers_sweep.sh
