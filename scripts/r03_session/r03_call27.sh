mkdir -p gpurun_out/r03
for s in 2 3 4; do
  export BASIC_SCAN_SLOTS=$s
  echo "== BASIC_SCAN_SLOTS=$s"
  KODAK_CFGS="basic:3 basic:4 basic:6" bash scripts/kodak_workers.sh gpurun_out/r03/slots$s > gpurun_out/r03/kodak_slots$s.log 2>&1; cat gpurun_out/r03/slots$s/kodak_workers/summary.txt
done
