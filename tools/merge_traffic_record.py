#!/usr/bin/env python3
"""Append / replace a PMC traffic record (tools/profile_round.sh -> gpurun_out/<tag>_traffic_record.json) in profiles/conv_hbm_traffic.json:
one record per (profile file, config, precision); usage: merge_traffic_record.py gpurun_out/r5_traffic_record.json [more records]"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
path = os.path.join(ROOT, "profiles", "conv_hbm_traffic.json")
recs = json.load(open(path))
for f in sys.argv[1:]:
    r = json.load(open(f))
    recs = [x for x in recs if not (x.get("profile") == r["profile"] and x.get("config") == r["config"] and x.get("precision") == r["precision"])]
    recs.append(r)
json.dump(recs, open(path, "w"), indent=1)
print("records:", len(recs), "last:", recs[-1])
