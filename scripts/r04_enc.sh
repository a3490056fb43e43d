#!/bin/bash
# after an encoder change: GPU suite, stand-alone chain timings, one-stream Kodak-shaped BaSIC item timeline
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/chain
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_enc2.log 2>&1; rc=$?
tail -3 $O/pytest_enc2.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/r04_rans_ns.py > $O/rans_ns_head.txt 2>&1; grep stream $O/rans_ns_head.txt
bash scripts/r04_kodak_timeline.sh > $O/kodak_tl.txt 2>&1; grep "rans_encode_fast_kernel<1>\|scanline_pipelined" $O/kodak_tl.txt | head -4
