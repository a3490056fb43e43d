#!/bin/bash
# where does the upload of the pcie_inclusive leg go?  kernel + memory-copy trace of bench.py --input host
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pcie
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/bench.py --input host --no-cpu-baseline --no-extra-legs --no-dominant > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-300
ls $O/trace/*
python3 - <<PY
import csv, glob
f = glob.glob("$O/trace/**/*memory_copy_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(len(rows), rows[0].keys())
big = [r for r in rows if int(r.get("Size", r.get("size", 0)) or 0) > 1e8] if ("Size" in rows[0] or "size" in rows[0]) else []
import statistics
durs = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Direction", r.get("Name", ""))) for r in rows)
print("longest copies (ms):", durs[-12:])
PY
cp $(find $O/trace -name "*memory_copy_trace.csv" | head -1) $O/memcpy.csv
cp $(find $O/trace -name "*kernel_trace.csv" | head -1) $O/kernels.csv
rm -rf $O/trace
