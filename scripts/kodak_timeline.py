#!/usr/bin/env python3
"""Where a Kodak-shaped item's compress / decompress time goes: from a rocprofv3 --kernel-trace CSV of
tools/run_benchmark.py, print per-kernel totals, the busy time per HIP stream, and (for the last `tail_ms` of the trace)
the kernels longer than 50 us with the idle gap on their stream before each."""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
tail_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 160.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def nm(r):
    return r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r), r.get("Stream_Id", "?") + "/q" + r.get("Queue_Id", "?")) for r in rows)
t_end = max(e[1] for e in ev)
tot = defaultdict(lambda: [0, 0, 0])
for s, e, n, st in ev:
    t = tot[n]; t[0] += 1; t[1] += e - s; t[2] = max(t[2], e - s)
print(f"{'kernel':60s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>9s} {'max_us':>9s}")
for n, (c, t, m) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{n[:60]:60s} {c:7d} {t / 1e6:9.2f} {t / c / 1e3:9.1f} {m / 1e3:9.1f}")
t0 = t_end - int(tail_ms * 1e6)
last = defaultdict(int)
print(f"\nlast {tail_ms:.0f} ms: kernels > 50 us (start ms, duration us, idle gap on the same stream before it us, stream, kernel)")
acc = 0
for s, e, n, st in ev:
    if s >= t0:
        acc += 1
        if e - s > 50000 or (last[st] and s - last[st] > 300000):
            print(f"{(s - t0) / 1e6:9.3f} {(e - s) / 1e3:9.0f} {(s - last[st]) / 1e3 if last[st] else 0:9.0f} {st:>6s} {n[:64]}")
    last[st] = max(last[st], e)
print(f"kernels in window: {acc}")

# persistent launches: how many are in flight when each starts, and every kernel that ran > 5 ms (possibly stalled)
pers = [(s, e, n, st) for s, e, n, st in ev if "scanline_p" in n]
print("\npersistent launches (start ms, duration ms, stream, in flight at start incl. itself) and other kernels > 5 ms:")
base = ev[0][0]
for s, e, n, st in ev:
    if "scanline_p" in n:
        inflight = sum(1 for s2, e2, _, _ in pers if s2 <= s < e2)
        print(f"{(s - base) / 1e6:10.2f} {(e - s) / 1e6:8.2f} {st:>6s} {inflight:2d}  {n[:50]}")
    elif e - s > 5000000:
        print(f"{(s - base) / 1e6:10.2f} {(e - s) / 1e6:8.2f} {st:>6s}     {n[:50]}")
