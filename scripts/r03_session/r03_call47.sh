mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x > gpurun_out/r03/pytest47.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r03/pytest47.log
[ $rc -eq 0 ] || { grep -E "^E  " gpurun_out/r03/pytest47.log | head -20; exit 1; }
for d in 0 1024 0 1024; do echo "BASIC_CONV_DEBUG=$d"; BASIC_CONV_DEBUG=$d timeout -k 10 120 python scripts/conv_layer_bench.py g_a.1 2>&1 | grep -v amdgpu; done
