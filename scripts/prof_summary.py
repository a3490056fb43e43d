#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals and the last N conv launches."""
import csv, glob, sys, collections
d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[name][0] += 1
    tot[name][1] += dur
allt = sum(v[1] for v in tot.values())
print(f"{'kernel':60s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'%':>6s}")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{k[:60]:60s} {v[0]:6d} {v[1]:12.0f} {v[1]/v[0]:10.1f} {100*v[1]/allt:6.2f}")
if n_last:
    convs = [r for r in rows if "conv" in r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]]
    for r in convs[-n_last:]:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-44:], r["Grid_Size_X"], "lds", r["LDS_Block_Size"], "vgpr", r["VGPR_Count"], r["Accum_VGPR_Count"], f"{dur:.0f} us")
