#!/bin/bash
# in-loop decoder profile on CODEC data (BaSIC level 0): 64 images of 256x256 (batched kernel), one Kodak-shaped image (pipelined kernel)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/chain
mkdir -p $O
cd $R
B=64 timeout -k 10 300 python scripts/r04_basic_scan_profile.py > $O/basic64.log 2>&1 || { echo "failed"; tail -5 $O/basic64.log; exit 1; }
grep "decode" $O/basic64.log | tail -2 | cut -c1-1200
B=1 SIZE=${SIZE:-512x768} timeout -k 10 300 python scripts/r04_basic_scan_profile.py > $O/basic1.log 2>&1 || { echo "failed"; tail -5 $O/basic1.log; exit 1; }
grep "decode" $O/basic1.log | tail -2 | cut -c1-1200
