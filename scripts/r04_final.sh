#!/bin/bash
# round-4 end-of-round validation (through gpurun): the driver's GPU tier (pytest -m gpu, smoke), then the default bench line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04final
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?
tail -3 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
s=$(date +%s)
timeout -k 10 600 python bench.py > $O/bench_line.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
echo "bench wall $(( $(date +%s) - s )) s"
python scripts/r04_show_line.py $O/bench_line.json
