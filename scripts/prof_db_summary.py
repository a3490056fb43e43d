#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 rocpd database (the default output of --kernel-trace): prof_db_summary.py results.db [last_n]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
name_col = "display_name" if "display_name" in cols else "kernel_name"
rows = list(cur.execute(f"select s.{name_col}, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
tot = collections.defaultdict(lambda: [0, 0.0])
clean = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
for n, s, e, *_ in rows:
    tot[clean(n)][0] += 1
    tot[clean(n)][1] += (e - s) / 1e3
allt = sum(v[1] for v in tot.values())
print(f"{'kernel':60s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'%':>6s}")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{k[:60]:60s} {v[0]:6d} {v[1]:12.0f} {v[1]/v[0]:10.1f} {100*v[1]/allt:6.2f}")
for n, s, e, gx, gy, wx in rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 0):] if len(sys.argv) > 2 else []:
    print(f"{clean(n)[-50:]:50s} grid {gx}x{gy} wg {wx}  {(e - s) / 1e3:9.1f} us")
