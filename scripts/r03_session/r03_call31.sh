mkdir -p gpurun_out/r03
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r03/pytest31.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 gpurun_out/r03/pytest31.log
[ $rc -eq 0 ] || exit 1
KODAK_CFGS="basic:0 basic:4 hyperprior:0 hyperprior:3" bash scripts/kodak_workers.sh gpurun_out/r03 > gpurun_out/r03/kodak_workers6.log 2>&1; cat gpurun_out/r03/kodak_workers/summary.txt
