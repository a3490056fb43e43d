mkdir -p gpurun_out/r03
for i in 1 2 3; do KODAK_CFGS="basic:0" bash scripts/kodak_workers.sh gpurun_out/r03 > gpurun_out/r03/kodak_workers7.log 2>&1; cat gpurun_out/r03/kodak_workers/summary.txt; done
