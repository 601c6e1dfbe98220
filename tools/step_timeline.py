#!/usr/bin/env python3
"""One step of a rocprofv3 --kernel-trace csv as a timeline: every kernel between two consecutive conformer_update_kernel launches with its start
offset, duration, queue and the idle time of the device before it.  usage: step_timeline.py <kernel_trace.csv> [--step N]"""
import argparse
import csv

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--step", type=int, default=20)
a = ap.parse_args()
rows = sorted(csv.DictReader(open(a.csv)), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "conformer_update_kernel" in r["Kernel_Name"]]
lo, hi = ends[a.step] + 1, ends[a.step + 1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
busy_until, total_busy = t0, 0
print(f"{'start us':>9s} {'dur us':>8s} {'idle before':>11s} q  kernel")
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    idle = max(0, s - busy_until) / 1e3
    total_busy += max(0, e - max(s, busy_until))
    busy_until = max(busy_until, e)
    name = r["Kernel_Name"].replace("void ", "").split("(")[0][:70]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {idle:11.1f} {r.get('Queue_Id', '?'):>2s} {name}")
print(f"step: {(busy_until - t0) / 1e3:.1f} us wall, {total_busy / 1e3:.1f} us with a kernel running, {hi - lo} launches")
