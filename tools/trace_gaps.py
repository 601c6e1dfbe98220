#!/usr/bin/env python3
"""GPU idle gaps in a rocprofv3 --kernel-trace db: prints every gap between consecutive kernel executions longer than --min-us and the
busy / idle totals of the window that starts at the first kernel whose name contains --after.
usage: trace_gaps.py <db> [--min-us 500] [--after randomize_draws]"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--min-us", type=float, default=500.0)
ap.add_argument("--after", default=None)
a = ap.parse_args()
cur = sqlite3.connect(a.db).cursor()
cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
namecol = "name" if "name" in cols else "kernel_name"
rows = list(cur.execute(f"select {namecol}, start, end from kernels order by start"))
if a.after:
    idx = next((i for i, r in enumerate(rows) if a.after in r[0]), 0)
    rows = rows[idx:]
t_end, busy, gaps = rows[0][1], 0.0, []
for n, s, e in rows:
    if s > t_end:
        g = (s - t_end) / 1e3
        if g >= a.min_us:
            gaps.append((round((s - rows[0][1]) / 1e6, 1), round(g / 1e3, 2), n[:40]))
    busy += max(0, e - max(s, t_end)) / 1e3
    t_end = max(t_end, e)
span = (t_end - rows[0][1]) / 1e3
print(f"window {span / 1e3:.1f} ms, busy {busy / 1e3:.1f} ms ({100 * busy / span:.1f} %), gaps >= {a.min_us} us: {len(gaps)}, their sum {sum(g[1] for g in gaps):.1f} ms")
for g in gaps:
    print("  at +%.1f ms: idle %.2f ms before %s" % g)
