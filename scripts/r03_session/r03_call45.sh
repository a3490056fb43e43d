mkdir -p gpurun_out/r03
BATCH=256 STEPS=24 OUT=gpurun_out/r03/sweep_b256.txt SWEEP="3:steps:8 3:steps:16 3:steps:4 4:steps:8 4:steps:16 2:steps:8 3:steps:8" bash scripts/workers_sweep.sh > /dev/null 2>&1
cat gpurun_out/r03/sweep_b256.txt
