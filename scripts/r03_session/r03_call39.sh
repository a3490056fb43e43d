mkdir -p gpurun_out/r03
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r03/pytest39.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r03/pytest39.log
[ $rc -eq 0 ] || exit 1
KODAK_CFGS="basic:0 basic:3 hyperprior:0" bash scripts/kodak_workers.sh gpurun_out/r03 > gpurun_out/r03/kodak_workers9.log 2>&1; cat gpurun_out/r03/kodak_workers/summary.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
