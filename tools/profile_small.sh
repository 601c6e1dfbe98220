#!/bin/bash
# kernel-level profile of a small batch (default 40 poses = one reference-style complex): where a latency-bound step goes
set -u
P=${1:-40}
cd "$(dirname "$0")/.." || exit 1
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf /tmp/prof_small; mkdir -p /tmp/prof_small
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_small/trace -o t -- python $OLDPWD/bench.py --poses $P --steps 40 --warmup 5 --no-alt --no-e2e --no-cpu-baseline --no-other ${CFG:+--config $CFG} > /tmp/prof_small/trace.out 2>&1 < /dev/null)
T=$(find /tmp/prof_small/trace -name '*.db' | head -1)
[ -n "$T" ] || { echo "no trace db"; tail -5 /tmp/prof_small/trace.out; exit 1; }
python tools/summarize_prof.py gpurun_out/small${P} "$T" < /dev/null
grep -v "^{" /tmp/prof_small/trace.out | tail -2 | cut -c1-200
grep "^{" /tmp/prof_small/trace.out | cut -c1-260
head -40 gpurun_out/small${P}_kernel_stats.csv
