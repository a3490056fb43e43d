set -x
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_scanline.py tests/test_gpu_pgm.py tests/test_gpu_ar_codecs.py tests/test_gpu_codec_graph.py -q -m gpu -x > gpurun_out/r03/pytest17.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -6 gpurun_out/r03/pytest17.log
timeout -k 10 300 python scripts/scanline_probe.py > gpurun_out/r03/scanline_probe5.txt 2>&1; cat gpurun_out/r03/scanline_probe5.txt
timeout -k 10 300 python bench.py --workload basic --no-cpu-baseline > gpurun_out/r03/bench_basic3.json 2> gpurun_out/r03/bench_basic3.err; head -c 400 gpurun_out/r03/bench_basic3.json; echo
