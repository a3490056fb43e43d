set -x
mkdir -p gpurun_out/r03
PROBE_PERM=1 timeout -k 10 300 python scripts/mconv_probe.py perm_wide: perm_nostage:BASIC_MCONV_DEBUG=1 perm_nostore:BASIC_MCONV_DEBUG=4 > gpurun_out/r03/mconv_probe5.txt 2>&1; cat gpurun_out/r03/mconv_probe5.txt
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_pgm.py tests/test_gpu_ar_codecs.py tests/test_gpu_scanline.py tests/test_gpu_codec_graph.py tests/test_gpu_reference_kats.py tests/test_gpu_codec.py -q -m gpu -x > gpurun_out/r03/pytest7.log 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/r03/pytest7.log
timeout -k 10 300 python bench.py --workload checkerboard --no-cpu-baseline > gpurun_out/r03/bench_cb4.json 2> gpurun_out/r03/bench_cb4.err; head -c 1300 gpurun_out/r03/bench_cb4.json; tail -3 gpurun_out/r03/bench_cb4.err
