#!/bin/bash
# kernel trace of bench.py --workload basic (through gpurun): per-kernel totals + the last 70 ms as a timeline
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04basic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W=${WORKERS:-3}
timeout -k 10 250 rocprofv3 --kernel-trace --stats -d $O/trace -o b --output-format csv -- python3 $R/bench.py --workload basic --workers $W --steps 9 --warmup 3 --no-cpu-baseline > $O/trace_w$W.log 2>&1 || { echo "trace failed"; tail -5 $O/trace_w$W.log; }
cd $R
python scripts/prof_summary.py gpurun_out/r04basic/trace > $O/summary_w$W.txt 2>&1
python scripts/timeline.py gpurun_out/r04basic/trace 70 > $O/timeline_w$W.txt 2>&1
rm -rf $O/trace
head -18 $O/summary_w$W.txt
