#!/bin/bash
# batched scan-line kernel: launch times by HIP events, then kernel durations by rocprofv3 (through gpurun)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/bprobe
mkdir -p $O
cd $R
timeout -k 10 200 python scripts/scan_batched_probe.py > $O/probe.log 2>&1 || { echo "probe failed"; tail -5 $O/probe.log; exit 1; }
cat $O/probe.log
cd /tmp && export TMPDIR=/tmp
PROBE=${PROF_SHAPE:-64x16x16} timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/trace -o p --output-format csv -- python3 $R/scripts/scan_batched_probe.py > $O/prof.log 2>&1 || { echo "rocprof failed"; tail -5 $O/prof.log; }
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 $f | cut -c1-220 > $O/kernel_stats_head.txt && cat $O/kernel_stats_head.txt
rm -rf $O/trace
