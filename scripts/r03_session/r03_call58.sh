mkdir -p gpurun_out/r03
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r03/pytest58.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r03/pytest58.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
