mkdir -p gpurun_out/r03
for kb in 112 126; do
  export BASIC_SCAN_WEIGHT_KB=$kb
  echo "== BASIC_SCAN_WEIGHT_KB=$kb"
  KODAK_CFGS="basic:0 basic:3 basic:4 basic:5" bash scripts/kodak_workers.sh gpurun_out/r03/wkb$kb > gpurun_out/r03/kodak_wkb$kb.log 2>&1; cat gpurun_out/r03/wkb$kb/kodak_workers/summary.txt
done
