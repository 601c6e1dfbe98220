#!/usr/bin/env python3
"""Print per-kernel averages of every counter in a rocprofv3 --pmc sqlite db (plus duration when dispatch times are present)."""
import collections
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
rows = list(cur.execute("select kernel_name, counter_name, value, start, end from counters_collection"))
acc = collections.defaultdict(list)
dur = collections.defaultdict(list)
for n, c, v, s, e in rows:
    key = n.replace("void ", "").split("(")[0][:50]
    acc[(key, c)].append(v)
    dur[key].append((e - s) / 1e3)
for (k, c), v in sorted(acc.items()):
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(f"{k},{c},{len(v)},{sum(v) / len(v):.1f},dur_us={sum(dur[k]) / len(dur[k]):.1f}")
