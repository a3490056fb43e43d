#!/bin/bash
# stand-alone rANS chains, ns per symbol: this tree's library, then (if present) the library built from the commit before
# the round-4 chain work (_old/, not tracked) swapped in for the same script
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/chain
mkdir -p $O
cd $R
timeout -k 10 300 python scripts/r04_rans_ns.py > $O/rans_ns_head.txt 2>&1 || { tail -5 $O/rans_ns_head.txt; exit 1; }
echo "== HEAD"; grep stream $O/rans_ns_head.txt
WIDE=1 timeout -k 10 300 python scripts/r04_rans_ns.py > $O/rans_ns_head_wide.txt 2>&1; echo "== HEAD, wide rows"; grep stream $O/rans_ns_head_wide.txt
if [ -f _old/cbench_basic_amd/libbasic_hip.so ]; then
  cp cbench_basic_amd/libbasic_hip.so /tmp/head_lib.so
  cp _old/cbench_basic_amd/libbasic_hip.so cbench_basic_amd/libbasic_hip.so
  timeout -k 10 300 python scripts/r04_rans_ns.py > $O/rans_ns_before.txt 2>&1 || tail -5 $O/rans_ns_before.txt
  WIDE=1 timeout -k 10 300 python scripts/r04_rans_ns.py > $O/rans_ns_before_wide.txt 2>&1
  cp /tmp/head_lib.so cbench_basic_amd/libbasic_hip.so
  echo "== before (5b0efb7)"; grep stream $O/rans_ns_before.txt; echo "== before, wide rows"; grep stream $O/rans_ns_before_wide.txt
fi
