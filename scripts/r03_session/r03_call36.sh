mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_reference_kats.py -q -m gpu -x -s -k tans > gpurun_out/r03/pytest36.log 2>&1; rc=$?; echo "pytest rc $rc"; grep -v amdgpu.ids gpurun_out/r03/pytest36.log | grep "tANS,\|passed\|failed\|Error\|error" | tail -12
