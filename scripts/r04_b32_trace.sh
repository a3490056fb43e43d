#!/bin/bash
# per-kernel times of one 32-image step, one stream (no overlap), vs 1/8 of the 256-image step (through gpurun)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b32
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 32 256; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/t$b -o t --output-format csv -- python3 $R/bench.py --batch $b --workers 1 --steps 8 --warmup 2 --no-cpu-baseline --no-extra-legs --no-dominant > $O/t$b.log 2>&1 || { echo "trace $b failed"; tail -3 $O/t$b.log; }
  cp $(find $O/t$b -name "*kernel_stats.csv" | head -1) $O/stats_b$b.csv
  rm -rf $O/t$b
done
cd $R
python - <<'PY'
import csv
def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        d[n] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3)
    return d
a, b = load("gpurun_out/r04b32/stats_b32.csv"), load("gpurun_out/r04b32/stats_b256.csv")
tot_a = tot_b = 0
print(f"{'kernel':58s} {'calls':>5s} {'b32 avg us':>10s} {'b256/8 us':>10s} {'ratio':>6s}")
for n, (c, avg, tot) in sorted(a.items(), key=lambda kv: -kv[1][2])[:28]:
    if n in b:
        ref = b[n][1] / 8 * (b[n][0] / c)
        print(f"{n[:58]:58s} {c:5d} {avg:10.1f} {ref:10.1f} {avg / ref if ref else 0:6.2f}")
        tot_a += tot; tot_b += b[n][2] / 8
print("total us (listed kernels): b32", round(tot_a), " b256/8", round(tot_b))
PY
