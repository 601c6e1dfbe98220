#!/bin/bash
# quick per-kernel time table of the default bench step: rocprofv3 --kernel-trace --stats -> gpurun_out/<tag>_stats.txt
tag=${1:-trace}; shift || true
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf /tmp/prof_$tag; mkdir -p /tmp/prof_$tag
R=$PWD
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o t -f csv -- python $R/bench.py --steps 40 --warmup 5 --no-alt --no-e2e --no-cpu-baseline --no-other "$@" > /tmp/prof_$tag/out.txt 2>&1)
S=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
python - "$S" > gpurun_out/${tag}_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print(f'{r["Name"][:100]:100s} {int(r["Calls"]):6d} {float(r["AverageNs"]) / 1e3:9.2f} us {100 * float(r["TotalDurationNs"]) / tot:6.2f} %')
PY
cat gpurun_out/${tag}_stats.txt; tail -c 400 /tmp/prof_$tag/out.txt
