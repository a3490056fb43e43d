#!/bin/bash
# wide-row layout change: the rANS / scan-line / KAT tests, then the stand-alone chain timings on narrow and wide rows
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/chain
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_wide.log 2>&1; rc=$?
tail -3 $O/pytest_wide.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/r04_rans_ns.py > $O/rans_ns_head.txt 2>&1; grep stream $O/rans_ns_head.txt
WIDE=1 timeout -k 10 300 python scripts/r04_rans_ns.py > $O/rans_ns_head_wide.txt 2>&1; echo "== wide rows"; grep stream $O/rans_ns_head_wide.txt
