#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace CSV: for the last `window_ms` of the trace, list kernels with start/end
relative times, queue / stream ids, and compute how much of each rANS kernel's interval is covered by MFMA conv kernels
running concurrently (overlap evidence)."""
import csv, glob, sys
d = sys.argv[1]
window_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 80.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def nm(r):
    return r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows))
# window = the last window_ms before the final rANS kernel ends (the timed loop; the roofline measurement passes that
# follow it in bench.py run on one stream), or before the end of the trace when there is no rANS kernel
rans_ends = [e[1] for e in ev if "rans_" in e[2]]
t_end = max(rans_ends) if rans_ends else max(e[1] for e in ev)
ev = [e for e in ev if e[0] <= t_end]
t0 = t_end - int(window_ms * 1e6)
sel = [e for e in ev if e[0] >= t0]
print(f"{'start_ms':>9s} {'dur_us':>8s} {'q':>3s} {'s':>3s} kernel")
for s, e, n, q, st in sel:
    if e - s > 20000:
        print(f"{(s - t0) / 1e6:9.3f} {(e - s) / 1e3:8.0f} {q:>3s} {st:>3s} {n[:70]}")
pairs = sorted({(q, st) for _, _, _, q, st in sel})
print("(queue, stream) pairs in window:", pairs)
convs = [(s, e) for s, e, n, _, _ in ev if "conv" in n]
tot_r = cov = 0
for s, e, n, _, _ in sel:
    if "rans_" in n and e - s > 200000:
        c = sum(max(0, min(e, ce) - max(s, cs)) for cs, ce in convs)
        tot_r += e - s
        cov += min(c, e - s)
print(f"rANS kernel time in window {tot_r / 1e6:.2f} ms, of which concurrent with a conv kernel: {cov / 1e6:.2f} ms ({100 * cov / max(1, tot_r):.0f} %)")
busy = sorted((s, e) for s, e, *_ in sel)
u, cur_s, cur_e = 0, None, None
for s, e in busy:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            u += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    u += cur_e - cur_s
print(f"window {window_ms:.0f} ms: some kernel running {u / 1e6:.2f} ms")
