set -x
mkdir -p gpurun_out/r03
timeout -k 10 400 python scripts/leg_order_probe.py 32x6x4 32x6x4 256x3x1 32x6x4 32x6x4 32x3x1 256x3x1 32x6x4 > gpurun_out/r03/leg_order.txt 2>&1; cat gpurun_out/r03/leg_order.txt
PROBE_EMPTY_CACHE=1 timeout -k 10 400 python scripts/leg_order_probe.py 32x6x4 256x3x1 32x6x4 256x6x1 32x6x4 > gpurun_out/r03/leg_order2.txt 2>&1; cat gpurun_out/r03/leg_order2.txt
