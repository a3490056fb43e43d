#!/bin/bash
# decoder-chain measurements: instruction-latency probe (one lone wavefront), then the in-loop decoder's profile slots
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/chain
mkdir -p $O
cd $R
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/r04_chain_probe.hip -o /tmp/chain_probe 2> $O/build.log || { tail -5 $O/build.log; exit 1; }
timeout -k 10 120 /tmp/chain_probe > $O/probe.txt 2>&1 || { echo "probe failed"; tail -5 $O/probe.txt; exit 1; }
cat $O/probe.txt
BASIC_SCAN_PROFILE=1 PROBE=${PROBE:-64x16x16,1x32x48,2x32x48} timeout -k 10 300 python scripts/scan_batched_probe.py > $O/scan.log 2>&1 || { echo "scan probe failed"; tail -5 $O/scan.log; exit 1; }
grep -v "^$" $O/scan.log | cut -c1-900 | tail -40
