#!/bin/bash
# 32 images per step on ONE stream: wall time per step against the sum of kernel durations (what the host and the
# synchronisation points add to a step's latency)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/b32host
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--batch 32 --steps 40 --warmup 6 --workers 1 --no-cpu-baseline --no-extra-legs --no-ar-workloads --no-dominant"
timeout -k 10 200 python3 $R/bench.py $A > $O/plain.json 2> $O/plain.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o b --output-format csv -- python3 $R/bench.py $A > $O/trace.log 2>&1
cd $R
python - $O <<'PY'
import csv, glob, json, sys
o = sys.argv[1]
d = json.loads([l for l in open(o + "/plain.json") if l.startswith("{")][-1])
print("one stream, 32 images: %.2f ms per step wall (%.1f Mpix/s)" % (d["ms_per_step"], d["value"]))
f = glob.glob(o + "/trace/**/*kernel_stats.csv", recursive=True)[0]
tot = sum(float(r["TotalDurationNs"]) for r in csv.DictReader(open(f)))
steps = 46 + 1   # timed + warm-up + the quiet comparison call
print("sum of kernel durations: %.2f ms per step (%d steps incl. warm-up and the quiet call)" % (tot / steps / 1e6, steps))
PY
rm -rf $O/trace
