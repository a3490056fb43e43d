#!/bin/bash
# after a decoder change: the GPU suite, then the in-loop decoder profile on codec data
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/chain
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
bash scripts/r04_chain2.sh
