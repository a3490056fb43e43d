mkdir -p gpurun_out/r03
R=$PWD
cd /tmp && export TMPDIR=/tmp
for w in 4; do
  rm -rf /tmp/kt$w
  (cd $R && timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$w -- python3 tools/run_benchmark.py --warmup --codec basic --synthetic 24 --height 512 --width 768 --batch-size 1 --workers $w --complexity-levels 0 --out /tmp/kt_out$w > /tmp/kt$w.json 2> /tmp/kt$w.err) || { echo "trace w$w failed"; tail -5 /tmp/kt$w.err; }
  python3 $R/scripts/kodak_timeline.py /tmp/kt$w 1 > $R/gpurun_out/r03/kodak_timeline_w$w.txt 2>&1
  python3 - /tmp/kt$w > $R/gpurun_out/r03/kodak_streams_w$w.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def nm(r): return r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r), r["Stream_Id"]) for r in rows)
base = ev[0][0]
per = defaultdict(list)
for s, e, n, st in ev: per[st].append((s, e, n))
for st, L in sorted(per.items()):
    print(f"== stream {st}: {len(L)} kernels")
    last_end = None
    run_start = None; run_busy = 0; run_n = 0
    for s, e, n in L:
        gap = (s - last_end) / 1e6 if last_end else 0
        if gap > 1.0 or e - s > 3e6:
            if run_n: print(f"      ... {run_n} short kernels, busy {run_busy / 1e6:.2f} ms")
            run_n = 0; run_busy = 0
            print(f"{(s - base) / 1e6:10.2f}  gap {gap:7.2f}  dur {(e - s) / 1e6:7.2f}  {n[:60]}")
        else:
            run_n += 1; run_busy += e - s
        last_end = max(last_end or 0, e)
PY
done
tail -5 /tmp/kt4.json | cut -c1-600
